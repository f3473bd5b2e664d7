"""Is a captured hipMemsetAsync honoured (and ordered) on every hipGraph replay?"""
import ctypes, torch
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
dev = torch.device("cuda:0")
for n in (1024, 76800 * 4 + 16, 1 << 20, (1 << 24) + 4, 147456 * 4):
    buf = torch.empty(n, dtype=torch.uint8, device=dev)
    acc = torch.zeros(16, device=dev)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(2):
            buf.fill_(255); hip.hipMemsetAsync(buf.data_ptr(), 0, n, s.cuda_stream); acc += 1
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    fails = 0
    with torch.cuda.graph(g):
        st = torch.cuda.current_stream().cuda_stream
        y = buf.float().sum()                       # reader before (forces ordering)
        hip.hipMemsetAsync(buf.data_ptr(), 0, n, st)
        z = buf.view(torch.int32)[: n // 4].abs().max()   # reader right after the memset
        buf.add_(1)                                 # dirty again for the next replay
    for rep in range(6):
        if rep % 2: buf.fill_(0xFF)
        g.replay(); torch.cuda.synchronize()
        if int(z) != 0: fails += 1
        print(f"n={n} replay {rep}: max after memset = {int(z)}  sum before = {float(y):.0f}", flush=True)
    print("n", n, "FAILS", fails)
