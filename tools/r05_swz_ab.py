"""Round 5: LDS chunk swizzle of the fp32 plane-product kernels, old ((row >> 2) & 3: 2-way conflict under the real
ds_read_b128 lane groups) against new (-(row >> 2) & 3: conflict-free).  The arithmetic is the same, so results must be
bitwise equal.  Needs both libraries:
    make -C psgd_tf_amd/csrc
    make -C psgd_tf_amd/csrc BUILD=$PWD/psgd_tf_amd/csrc/build_oldswz LIB=$PWD/psgd_tf_amd/csrc/build_oldswz/libpsgd_hip.so EXTRA_HIPFLAGS=-DPSGD_LDS_SWZ_OLD
    python tools/r05_swz_ab.py            (parent: runs itself once per library, compares digests and times)
"""
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = ((4096, 4096), (2048, 4096), (1000, 3000), (6144, 6144))


def worker():
    import torch
    sys.path.insert(0, ROOT)
    import preconditioned_stochastic_gradient_descent as psgd
    from tools.kron_bf16_update_timing import tri, timeit
    out = {}
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    for M, N in SHAPES:
        Ql, Qr = tri(M, g), tri(N, g)
        dX = torch.randn(M, N, device="cuda", generator=g)
        dG = dX * 1.5 + 0.1 * torch.randn(M, N, device="cuda", generator=g)
        G = torch.randn(M, N, device="cuda", generator=g)
        dig = hashlib.sha256()
        new = psgd.update_precond_kron(Ql, Qr, dX, dG, 0.01)
        for t in new:
            dig.update(t.cpu().numpy().tobytes())
        pg = psgd.precond_grad_kron(Ql.clone(), Qr.clone(), G)        # first sight: the Gram-free chain
        dig.update(pg.cpu().numpy().tobytes())
        nb = psgd.update_precond_kron(Ql, Qr, dX.bfloat16(), dG.bfloat16(), 0.01)   # (fp32 solves inside the bf16-operand update)
        for t in nb:
            dig.update(t.cpu().numpy().tobytes())
        t_up = min(timeit(lambda: psgd.update_precond_kron(Ql, Qr, dX, dG, 0.01), 8) for _ in range(3))
        t_ub = min(timeit(lambda: psgd.update_precond_kron(Ql, Qr, dX.bfloat16(), dG.bfloat16(), 0.01), 8) for _ in range(3))

        def apply_new():
            return psgd.precond_grad_kron(Ql.clone(), Qr.clone(), G)
        t_ap = min(timeit(apply_new, 8) for _ in range(3))
        out["%dx%d" % (M, N)] = {"sha256": dig.hexdigest(), "update_ms": t_up, "update_bf16ops_ms": t_ub, "apply_new_factors_ms": t_ap}
    print("RESULT " + json.dumps(out))


def main():
    libs = {"old": os.path.join(ROOT, "psgd_tf_amd/csrc/build_oldswz/libpsgd_hip.so"),
            "new": os.path.join(ROOT, "psgd_tf_amd/csrc/libpsgd_hip.so")}
    res = {}
    for rnd in range(2):
        for tag, path in libs.items():
            env = dict(os.environ, PSGD_HIP_LIB=path)
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "worker"], env=env, capture_output=True, text=True)
            line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")]
            if not line:
                print(tag, "FAILED", p.stdout[-2000:], p.stderr[-2000:]); return 1
            r = json.loads(line[0][7:])
            if tag not in res:
                res[tag] = r
            else:
                for s in r:
                    assert r[s]["sha256"] == res[tag][s]["sha256"], "run-to-run bits differ"
                    for k in r[s]:
                        if k.endswith("_ms"):
                            res[tag][s][k] = min(res[tag][s][k], r[s][k])
    ok = True
    for s in res["old"]:
        o, n = res["old"][s], res["new"][s]
        same = o["sha256"] == n["sha256"]
        ok &= same
        print("%-10s bitwise equal %s | fp32 update %.3f -> %.3f ms | bf16-operand update %.3f -> %.3f ms | fp32 apply (new factors) %.3f -> %.3f ms"
              % (s, same, o["update_ms"], n["update_ms"], o["update_bf16ops_ms"], n["update_bf16ops_ms"],
                 o["apply_new_factors_ms"], n["apply_new_factors_ms"]))
    return 0 if ok else 1


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "worker":
        worker()
    else:
        sys.exit(main())
