import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
print("torch cuda available", torch.cuda.is_available(), torch.cuda.device_count())
from ntt_aie_amd import _lib
L = _lib.lib()
print("maps:", sorted(set(l.split()[-1] for l in open('/proc/self/maps') if 'amdhip' in l or 'hsa-runtime' in l)))
print("ntt_device_count", L.ntt_device_count())
hip = C.CDLL("libamdhip64.so.7")
n = C.c_int(-1)
print("hipGetDeviceCount rc", hip.hipGetDeviceCount(C.byref(n)), n.value)
x = torch.zeros(4, device="cuda")
print("after torch init: ntt_device_count", L.ntt_device_count())
h = C.c_void_p()
print("plan_create", L.ntt_plan_create(C.byref(h), 8, 3329, 4, 0))
for k in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "LD_LIBRARY_PATH", "LD_PRELOAD"):
    print(k, os.environ.get(k))
