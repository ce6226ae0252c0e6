"""Host <-> device copy rates of this box (page-locked host memory, hipMemcpyAsync): one copy, the same bytes cut over 2 / 4 streams,
and both directions at once -- what bounds bench.py --host-images (157 MB up, 38 MB down per step)."""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
H = C.CDLL("libamdhip64.so")
vp = C.c_void_p
H.hipMemcpyAsync.argtypes = [vp, vp, C.c_size_t, C.c_int, vp]
H.hipStreamSynchronize.argtypes = [vp]
H.hipStreamCreateWithFlags.argtypes = [C.POINTER(vp), C.c_uint]
n = 160 << 20
h = torch.empty(n, dtype=torch.uint8).pin_memory(); d = torch.empty(n, dtype=torch.uint8, device="cuda")
h2 = torch.empty(n // 4, dtype=torch.uint8).pin_memory(); d2 = torch.empty(n // 4, dtype=torch.uint8, device="cuda")
qs = []
for _ in range(5):
    q = vp(); H.hipStreamCreateWithFlags(C.byref(q), 1); qs.append(q)
def run(parts, down=False, both=False, reps=10):
    def once():
        step = n // parts
        for i in range(parts):
            if down:
                H.hipMemcpyAsync(h.data_ptr() + i * step, d.data_ptr() + i * step, step, 2, qs[i])
            else:
                H.hipMemcpyAsync(d.data_ptr() + i * step, h.data_ptr() + i * step, step, 1, qs[i])
        if both:
            H.hipMemcpyAsync(h2.data_ptr(), d2.data_ptr(), n // 4, 2, qs[4])
        for q in qs: H.hipStreamSynchronize(q)
    once(); torch.cuda.synchronize()
    a = time.perf_counter()
    for _ in range(reps): once()
    dt = (time.perf_counter() - a) / reps
    return (n + (n // 4 if both else 0)) / dt / 1e9, dt * 1e3
for parts in (1, 2, 4):
    print("H2D 160 MiB over %d stream(s): %.1f GB/s (%.2f ms)" % ((parts,) + run(parts)))
for parts in (1, 2):
    print("D2H 160 MiB over %d stream(s): %.1f GB/s (%.2f ms)" % ((parts,) + run(parts, down=True)))
for parts in (1, 2, 4):
    print("H2D 160 MiB over %d stream(s) + D2H 40 MiB beside it: %.1f GB/s in total (%.2f ms)" % ((parts,) + run(parts, both=True)))
