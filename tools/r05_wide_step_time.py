"""The UVd.step pattern (update, then apply on the updated state) at ranks 33 .. 64: psgd_uvd_wide_update_apply_f32 against the two
separate calls (PSGD_WIDE_STEP=0), N rows.   python tools/r05_wide_step_time.py [N]"""
import os
import subprocess
import sys
import torch

if len(sys.argv) > 2 and sys.argv[2] == "child":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import preconditioned_stochastic_gradient_descent as psgd
    N = int(sys.argv[1])
    dev = torch.device("cuda")
    for r in (32, 40, 48, 64):
        g = torch.Generator(device=dev).manual_seed(r)
        sc = (1.0 / (N * r)) ** 0.5
        U, V = torch.randn(N, r, device=dev, generator=g) * sc, torch.randn(N, r, device=dev, generator=g) * sc
        d = torch.ones(N, 1, device=dev)
        gr, v = torch.randn(N, 1, device=dev, generator=g), torch.randn(N, 1, device=dev, generator=g)
        h = v * 1.5
        flip = [0]

        def step():
            flip[0] ^= 1
            return psgd.update_precond_UVd_math_and_precond_grad(U, V, d, v, h, gr, 0.01, 1e-38, balance=False, update_U=bool(flip[0]))
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(8):
            step()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 8
        print("r=%2d  step %6.2f ms  %5.2f TB/s on 4(9r+15) B/row   [%s]" % (r, ms, 4 * (9 * r + 15) * N / ms / 1e9,
                                                                            "two calls" if os.environ.get("PSGD_WIDE_STEP") == "0" else "fused"))
        del U, V
    sys.exit(0)

N = sys.argv[1] if len(sys.argv) > 1 else "20000000"
for env in ({}, {"PSGD_WIDE_STEP": "0"}):
    r = subprocess.run([sys.executable, __file__, N, "child"], env=dict(os.environ, **env), capture_output=True, text=True)
    print(r.stdout.strip() if r.returncode == 0 else (r.stdout + r.stderr[-3000:]))
