import ctypes, os, sys
order = sys.argv[1]
def load():
    L = ctypes.CDLL("halo2-plonky2-verifier_amd/libh2w.so")
    L.h2w_device_count.restype = ctypes.c_int
    return L
if order == "torch_first":
    import torch
    print("torch avail", torch.cuda.is_available(), torch.version.hip)
    L = load(); print("h2w_device_count", L.h2w_device_count())
    x = torch.zeros(4, device="cuda"); print("torch alloc ok")
    print("h2w_device_count after", L.h2w_device_count())
elif order == "lib_first":
    L = load(); print("h2w_device_count", L.h2w_device_count())
    import torch
    print("torch avail", torch.cuda.is_available())
    x = torch.zeros(4, device="cuda"); print("torch alloc ok", L.h2w_device_count())
maps = open("/proc/self/maps").read()
print(sorted({l.split()[-1] for l in maps.splitlines() if "libamdhip64" in l or "libhsa-runtime" in l}))
