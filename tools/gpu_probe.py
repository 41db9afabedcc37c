import ctypes as C, os, sys
sys.path.insert(0, os.getcwd())
order = sys.argv[1] if len(sys.argv) > 1 else "torch_first"
if order == "torch_first":
    import torch; print("torch avail", torch.cuda.is_available(), torch.version.hip)
import tiger_hlm_gpu_amd as H
L = H._lib.lib()
n = C.c_int(0); rc = L.hlm_device_count(C.byref(n)); print("rc", rc, "n", n.value, L.hlm_last_error())
if order != "torch_first":
    import torch; print("torch avail", torch.cuda.is_available())
os.system("grep -E 'amdhip|hsa-runtime' /proc/%d/maps | awk '{print $6}' | sort -u" % os.getpid())
