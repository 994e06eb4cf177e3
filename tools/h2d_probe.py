import torch, time, numpy as np
dev = "cuda:0"
a = torch.rand(8192, 8192, device=dev)
small = np.arange(30000, dtype=np.int32)
torch.cuda.synchronize()
def busy():
    for _ in range(3): (a @ a)
for mode in ("pageable", "pinned_ring"):
    pin = torch.empty(small.nbytes, dtype=torch.uint8, pin_memory=True)
    torch.cuda.synchronize()
    busy(); 
    t0 = time.perf_counter()
    if mode == "pageable":
        x = torch.from_numpy(small).to(dev)
    else:
        pin.copy_(torch.from_numpy(small.view(np.uint8)))
        x = torch.empty(30000, dtype=torch.int32, device=dev)
        x.view(torch.uint8).copy_(pin, non_blocking=True)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(mode, "host blocked %.3f ms; queued work finished after %.3f ms more; ok=%s" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, bool((x.cpu().numpy() == small).all())))
t0 = time.perf_counter(); p = torch.empty(200000, dtype=torch.uint8, pin_memory=True); t1 = time.perf_counter()
print("torch.empty(pin_memory=True) 200 KB: %.3f ms" % ((t1 - t0) * 1e3))
t0 = time.perf_counter(); p = torch.empty(200000, dtype=torch.uint8, pin_memory=True); t1 = time.perf_counter()
print("again: %.3f ms" % ((t1 - t0) * 1e3))
