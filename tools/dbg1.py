import ctypes as C, re, os, sys
sys.path.insert(0, os.getcwd())
import torch
print("torch cuda", torch.cuda.is_available(), torch.version.hip)
x = torch.zeros(4, device="cuda")
from gpudrive_lab_amd import _capi
L = _capi.lib()
m = open('/proc/self/maps').read()
print(sorted(set(re.findall(r'/\S*(?:amdhip64|hsa-runtime|rocprofiler-register|amd_comgr)\S*', m))))
hip = C.CDLL("libamdhip64.so.7")
n = C.c_int(-1)
print("hipGetDeviceCount via soname:", hip.hipGetDeviceCount(C.byref(n)), n.value)
rv = C.c_int(0); hip.hipRuntimeGetVersion(C.byref(rv)); print("runtime version", rv.value)
import subprocess
print(subprocess.run(["ldd", _capi.lib_path()], capture_output=True, text=True).stdout)
print({k:v for k,v in os.environ.items() if 'HIP' in k or 'ROC' in k or 'HSA' in k or 'LD_' in k})
