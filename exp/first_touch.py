"""does a throw-away allocate / fill / free cycle in the same (or an earlier) process change the step time of the first bench
process on a fresh box?  run on the GPU box: python exp/first_touch.py [gb]"""
import sys, time, torch
gb = int(sys.argv[1]) if len(sys.argv) > 1 else 250
t = time.time()
x = torch.empty((gb << 30) // 8, dtype=torch.int64, device="cuda")
x.fill_(-1)
torch.cuda.synchronize()
print("alloc+fill", gb, "GB:", round(time.time() - t, 2), "s")
t = time.time()
x.zero_()
torch.cuda.synchronize()
print("second fill:", round(time.time() - t, 3), "s")
del x
torch.cuda.empty_cache()
