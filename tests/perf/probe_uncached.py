"""Experiment: put the matrix stream (col_idx / values) in uncached or fine-grained device memory
(hipExtMallocWithFlags) so that it does not displace x from the XCD L2s; measure read rate and SpMV."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from loops_amd import generate as G, spmv as S, probes as PR, _lib

hip = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
hip.hipExtMallocWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]

class Raw:  # minimal tensor stand-in: device pointer + size
    def __init__(self, host, flags):
        self.host = np.ascontiguousarray(host); self.n = self.host.size
        self.p = C.c_void_p()
        rc = hip.hipExtMallocWithFlags(C.byref(self.p), self.host.nbytes, flags); assert rc == 0, rc
        rc = hip.hipMemcpy(self.p, self.host.ctypes.data_as(C.c_void_p), self.host.nbytes, 1); assert rc == 0, rc
        self.dtype = torch.float32 if self.host.dtype == np.float32 else torch.int32
        self.is_cuda = True
    def data_ptr(self): return self.p.value
    def numel(self): return self.n
    def is_contiguous(self): return True

def ev(fn, iters=30):
    for _ in range(3): fn()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in evs]))

torch.zeros(1).cuda()
rows = cols = 1 << 20; nnz = 1 << 24
deg = G.powerlaw_degrees(rows, nnz)
off, idx, val = G.powerlaw_csr(rows, cols, nnz, degrees=deg)
xh = G.uniform_distribution_int(cols)
x = torch.from_numpy(xh).cuda(); y = torch.empty(rows, device="cuda")
from oracle import oracle as O
ref = O.spmv_f32(off, idx, val, xh, omp=True)
offd = torch.from_numpy(off).cuda()
for name, flags in (("coarse (hipMalloc default)", None), ("fine-grained", 1), ("uncached", 3)):
    if flags is None:
        ci, cv = torch.from_numpy(idx).cuda(), torch.from_numpy(val).cuda()
    else:
        ci, cv = Raw(idx, flags), Raw(val, flags)
    csr = S.CSR(rows, cols, offd, ci, cv)
    csr.check = lambda *a: None
    plan = S.MergePathPlan(csr)
    ms = ev(lambda: S.merge_path_flat_stage(csr, x, y, plan, 0))
    S.merge_path_flat(csr, x, y, plan=plan); torch.cuda.synchronize()
    ok = np.array_equal(y.cpu().numpy(), ref)
    # read-only stream rate over the values array
    L = PR.lib()
    def rd():
        _lib.check(L.loops_stream_copy_f32(C.c_void_p(cv.data_ptr()), C.c_void_p(cv.data_ptr()), cv.numel(), C.c_void_p(torch.cuda.current_stream().cuda_stream)), "read")
    msr = ev(rd)
    print(f"{name:28s} fused kernel {ms*1e3:7.1f} us  exact={ok}  read-only stream of values {cv.numel()*4/msr/1e6:7.1f} GB/s", flush=True)
