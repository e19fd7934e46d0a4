import time, torch
side = torch.cuda.Stream()
def t(f, n=2000):
    for _ in range(50): f()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): f()
    dt=(time.perf_counter()-t0)/n*1e6; torch.cuda.synchronize(); return dt
def ctx():
    with torch.cuda.stream(side): pass
def ev():
    e = torch.cuda.Event(); e.record(side); torch.cuda.current_stream().wait_event(e)
x = torch.zeros(4, device="cuda")
print("stream ctx %.1f us" % t(ctx))
print("event create+record+wait %.1f us" % t(ev))
print("current_stream %.1f us" % t(lambda: torch.cuda.current_stream()))
print("randint %.1f us" % t(lambda: torch.randint(0, 384, (10,), device="cuda")))
print("torch.empty %.1f us" % t(lambda: torch.empty((1<<20,), dtype=torch.uint8, device="cuda")))
print("record_stream %.1f us" % t(lambda: x.record_stream(side)))
print("torch.cuda.device ctx %.1f us" % t(lambda: torch.cuda.device(0).__enter__()))
