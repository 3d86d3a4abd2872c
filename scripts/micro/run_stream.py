import ctypes, os, sys, torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "libstream_rec.so"))
lib.stream_rec.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
dev = "cuda:0"
for B in (4096, 8192):
    S = 50
    rec = torch.randn((B, S, 416), device=dev); out = torch.empty((B, 64), device=dev)
    big = torch.zeros(512 << 20, dtype=torch.uint8, device=dev)
    for pf, flush in ((1, True), (3, True), (6, True), (3, False)):
        ts = []
        for i in range(8):
            if flush: big.sum()            # flush the caches between runs with READS (a write flush leaves dirty lines to evict)
            torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); lib.stream_rec(rec.data_ptr(), B, S, pf, out.data_ptr(), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)); e1.record()
            torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
        t = sorted(ts)[len(ts) // 2] * 1e-3
        print(f"B={B} flush={flush} prefetch depth {pf}: {t*1e6:.1f} us  {B*S*1664/t/1e12:.2f} TB/s")
