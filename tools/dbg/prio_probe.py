import ctypes as C, torch
hip = C.CDLL("libamdhip64.so")
lo, hi = C.c_int(), C.c_int()
print("rc", hip.hipDeviceGetStreamPriorityRange(C.byref(lo), C.byref(hi)), "least", lo.value, "greatest", hi.value)
torch.cuda.init()
for p in (-2, -1, 0, 1, 2):
    try:
        s = torch.cuda.Stream(priority=p); print("torch stream priority", p, "->", s.priority)
    except Exception as e:
        print("torch stream priority", p, "error", e)
