"""256 x 288 large-K GEMM (csrc/gemm_wide2.hip, variants 10 / 11) against the shipped kernel and the 256 x 192 lock-step form."""
import os, sys, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from underwater_image_restoration_amd import ops, _lib
lib = _lib.load()
raw = ctypes.CDLL(os.path.join(os.path.dirname(_lib.__file__), "libuwir_hip.so"))
raw.uwir_gw2_run.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 3 + [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
raw.uwir_gw2_run.restype = ctypes.c_int

def timeit(f, reps=30):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

a = torch.randn(8192, 8192, device="cuda").bfloat16()
for _ in range(50): a @ a
torch.cuda.synchronize()
st = torch.cuda.current_stream().cuda_stream
variants = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0, 10, 11]
for M, N, K in ((262144, 576, 1728), (65536, 1152, 3456), (16384, 2304, 6912), (262144 - 100, 576, 1728)):
    A = torch.randn(M, K, device="cuda").bfloat16(); W = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16()
    bias = torch.randn(N, device="cuda")
    o1 = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    f1 = lambda: ops.conv_gemm(A, W, o1, B=1, H=1, W=M, C_=K, N=N, bias=bias)
    outs = {v: torch.zeros_like(o1) for v in variants}
    def mk(v):
        def f():
            rc = raw.uwir_gw2_run(A.data_ptr(), W.data_ptr(), outs[v].data_ptr(), bias.data_ptr(), M, N, K, None, v, st)
            assert rc == 0, lib.uwir_last_error()
        return f
    fs = {v: mk(v) for v in variants}
    res = {v: [] for v in variants}; r1 = []
    for _ in range(3):
        r1.append(timeit(f1))
        for v in variants: res[v].append(timeit(fs[v]))
    fl = 2.0 * M * N * K
    print(f"M{M} N{N} K{K}: shipped " + " ".join(f"{t:.3f}" for t in r1) + f" ms ({fl / min(r1) / 1e9:.0f} TF/s)")
    for v in variants:
        print(f"    v{v}: " + " ".join(f"{t:.3f}" for t in res[v]) + f" ms ({fl / min(res[v]) / 1e9:.0f} TF/s)  max |v - shipped| "
              f"{float((o1.float() - outs[v].float()).abs().max()):.3e}", flush=True)
