"""Regenerates the golden cases of this directory FROM THE REFERENCE ITSELF, under TensorFlow.

    python tests/golden/make_golden_tf.py [--reference /root/reference] [--out tests/golden/tf]

Build-container tool, never shipped to the GPU box and never imported by tests or the product: it
imports /root/reference/preconditioned_stochastic_gradient_descent.py (which needs `import tensorflow`,
psgd.py:18), feeds it the INPUTS of the committed oracle-made fixtures (tests/golden/*.npz: same seeds,
same arrays) and writes the reference's own fp32 outputs to tests/golden/tf/<case>.npz together with the
TensorFlow version.  tests/test_golden.py::test_oracle_matches_reference_fixtures then compares the
oracle with those files; while the directory is empty it reports PARITY UNPINNED.

Covered: UVd math (4 cases, both factor branches, the balance branch), one whole UVd.step per branch (uvdstep_*),
Kron dense (x) dense (4) and the six sparse dispatch formats (kron_fmt_*, psgd.py:198-391), sparse LU (3), and the
dense 2 x 2 first step of hello_psgd.py.

STATUS: TensorFlow is not installable in the build container of rounds 1-3 (no wheel, no network), so
this script has not run and tests/golden/tf/ is empty: parity is unpinned.  The
fixtures are data (inputs + the reference's outputs); no reference source is copied anywhere.

The reference draws its two branch decisions from tf.random.uniform([]) (psgd.py:562, :588).  To drive
both branches deterministically the script substitutes that one call, for the duration of one
update_precond_UVd_math_ call, by a function that hands out pre-chosen numbers (0.0 -> branch taken,
1.0 -> not taken); nothing else of TensorFlow or of the reference is touched.
"""
import argparse
import glob
import importlib.util
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def load_reference(ref_dir):
    path = os.path.join(ref_dir, "preconditioned_stochastic_gradient_descent.py")
    spec = importlib.util.spec_from_file_location("psgd_reference", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)           # raises ModuleNotFoundError here when TensorFlow is absent
    return mod


class Coins:
    """tf.random.uniform([]) replacement for one call: returns the queued scalars in order."""

    def __init__(self, tf, values):
        self.tf, self.values, self.saved = tf, list(values), None

    def __enter__(self):
        self.saved = self.tf.random.uniform
        tf, values = self.tf, self.values

        def uniform(shape, *a, **k):
            assert list(shape) == [] and values, "unexpected random draw in the reference"
            return tf.constant(values.pop(0), dtype=tf.float32)
        self.tf.random.uniform = uniform
        return self

    def __exit__(self, *exc):
        self.tf.random.uniform = self.saved
        assert not self.values, "the reference drew fewer numbers than expected"


def uvd_case(ref, tf, z):
    U, V, d = (tf.Variable(z[k]) for k in ("U", "V", "d"))
    g, v, h = (tf.constant(z[k]) for k in ("g", "v", "h"))
    pre0 = ref.precond_grad_UVd_math(U, V, d, g).numpy()
    coins = [0.0 if bool(z["balance"]) else 1.0, 0.0 if bool(z["update_U"]) else 1.0]      # psgd.py:562 then :588
    with Coins(tf, coins):
        ref.update_precond_UVd_math_(U, V, d, v, h, tf.constant(float(z["step"]), tf.float32),
                                     tf.constant(float(z["tiny"]), tf.float32))
    pre1 = ref.precond_grad_UVd_math(U, V, d, g).numpy()
    return dict(pre_grad_before=pre0, U_new=U.numpy(), V_new=V.numpy(), d_new=d.numpy(), pre_grad_after=pre1)


def kron_case(ref, tf, z):
    c = lambda k: tf.constant(z[k])
    Ql_new, Qr_new = ref.update_precond_kron(c("Ql"), c("Qr"), c("dX"), c("dG"), tf.constant(float(z["step"]), tf.float32))
    pre = ref.precond_grad_kron(c("Ql"), c("Qr"), c("G"))
    return dict(Ql_new=Ql_new.numpy(), Qr_new=Qr_new.numpy(), pre_grad=pre.numpy())


def splu_case(ref, tf, z):
    c = lambda k: tf.constant(z[k])
    pre = ref.precond_grad_splu(c("L12"), c("l3"), c("U12"), c("u3"), [c("g")])[0]
    new = ref.update_precond_splu(c("L12"), c("l3"), c("U12"), c("u3"), [c("dx")], [c("dg")], float(z["step"]))
    return dict(pre_grad=pre.numpy(), L12_new=new[0].numpy(), l3_new=new[1].numpy(), U12_new=new[2].numpy(),
                u3_new=new[3].numpy())


class Normals:
    """tf.random.normal replacement for one UVd.step call: hands out the fixture's probe vectors, one per parameter,
    in parameter order (psgd.py:713 draws them with tf.random.normal(param.shape))."""

    def __init__(self, tf, arrays):
        self.tf, self.arrays, self.saved = tf, list(arrays), None

    def __enter__(self):
        self.saved = self.tf.random.normal
        tf, arrays = self.tf, self.arrays

        def normal(shape, *a, **k):
            x = arrays.pop(0)
            assert tuple(shape) == tuple(x.shape), "unexpected probe-vector shape in the reference"
            return tf.constant(x)
        self.tf.random.normal = normal
        return self

    def __exit__(self, *exc):
        self.tf.random.normal = self.saved
        assert not self.arrays, "the reference drew fewer probe vectors than expected"


UVD_STEP_SHAPES = [(2, 30), (30, 30), (30,), (30, 1), (1,)]        # rnn_xor_UVd_preconditioner.py:28-31


def uvd_step_case(ref, tf, z):
    """One UVd.step of the reference itself (psgd.py:692-764) on the fixture's inputs: the closure is the separable
    loss sum(c p^2 / 2 + b p + e p^4 / 4) (TensorFlow differentiates it twice on its own), the probe vectors and the
    three coin flips (:703 update, :562 balance, :588 which factor) are the fixture's."""
    def unfl(x):
        out, i = [], 0
        for s in UVD_STEP_SHAPES:
            n = int(np.prod(s))
            out.append(np.reshape(x[i:i + n], s))
            i += n
        return out
    params = [tf.Variable(x) for x in unfl(z["p"])]
    cs, bs, es = ([tf.constant(x) for x in unfl(z[k])] for k in ("c", "b", "e"))
    max_norm = float(z["grad_clip_max_norm"])
    opt = ref.UVd(params, rank_of_modification=int(z["rank"]), preconditioner_init_scale=1.0,
                  lr_params=float(z["lr_params"]), lr_preconditioner=float(z["lr_preconditioner"]),
                  grad_clip_max_norm=(None if np.isinf(max_norm) else max_norm),
                  preconditioner_update_probability=1.0, exact_hessian_vector_product=True)
    opt._U.assign(z["U"]); opt._V.assign(z["V"]); opt._d.assign(z["d"])

    def closure():
        return tf.add_n([tf.reduce_sum(0.5 * c * p * p + b * p + 0.25 * e * p * p * p * p)
                         for p, c, b, e in zip(params, cs, bs, es)])
    coins = [0.0, 0.0 if bool(z["balance"]) else 1.0, 0.0 if bool(z["update_U"]) else 1.0]
    with Coins(tf, coins), Normals(tf, unfl(z["vs"])):
        loss = opt.step(closure)
    p_new = np.concatenate([np.reshape(p.numpy(), [-1]) for p in params], 0)
    return dict(loss=np.asarray(loss.numpy()), p_new=p_new, U_new=opt._U.numpy(), V_new=opt._V.numpy(), d_new=opt._d.numpy())


def dense_case(ref, tf):
    """hello_psgd.py:7-12,25-26 first iteration with v = (1, 0) (KAT-R of SURVEY Appendix C)."""
    Q = tf.constant(0.1 * np.eye(2, dtype=np.float32))
    dxs = [tf.constant(1.0), tf.constant(0.0)]
    dgs = [tf.constant(802.0), tf.constant(400.0)]
    Qn = ref.update_precond_dense(Q, dxs, dgs, step=0.2)
    pre = ref.precond_grad_dense(Qn, [tf.constant(-4.0), tf.constant(0.0)])
    return dict(Q_new=Qn.numpy(), pre_grad=np.array([p.numpy() for p in pre], dtype=np.float32))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default=os.environ.get("PSGD_REFERENCE_DIR", "/root/reference"))
    ap.add_argument("--out", default=os.path.join(HERE, "tf"))
    args = ap.parse_args()
    try:
        import tensorflow as tf
    except ModuleNotFoundError:
        sys.exit("make_golden_tf.py: TensorFlow is not importable here -- the reference cannot run, parity stays UNPINNED")
    ref = load_reference(args.reference)
    os.makedirs(args.out, exist_ok=True)
    meta = dict(tf_version=tf.__version__, generator="tests/golden/make_golden_tf.py")
    for path in sorted(glob.glob(os.path.join(HERE, "*.npz"))):
        name = os.path.basename(path)
        z = np.load(path)
        fn = (uvd_case if name.startswith("uvd_") else uvd_step_case if name.startswith("uvdstep_") else
              kron_case if name.startswith("kron_") else splu_case)      # (kron_fmt_*: the six sparse dispatch formats)
        np.savez_compressed(os.path.join(args.out, name), **fn(ref, tf, z), **meta)
        print("wrote", name)
    np.savez_compressed(os.path.join(args.out, "dense_hello_first_step.npz"), **dense_case(ref, tf), **meta)
    print("reference fixtures written to", args.out, "with TensorFlow", tf.__version__)


if __name__ == "__main__":
    main()
