"""One-shot check of the LDS-DMA weight path of conv3x3_halo_kernel (csrc/igemm.hip, build with -DMGU_DIAG=50): every weight tile,
as it stands in LDS behind the hand-counted vmcnt wait and the raw barrier, is compared with its global source before the step's
first MFMA.  Run with MGU_LIB_PATH=<the diag build>; prints the number of 16-byte pieces compared and the mismatch flag."""
import ctypes as C
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "mingraph-unet_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))
import mgunet
import mgunet_oracle as O
from mgunet import _lib
cuda = torch.device("cuda:0")
L = _lib.lib()
L.mgu_diag_glds_read.restype = C.c_int
L.mgu_diag_glds_read.argtypes = [C.POINTER(C.c_uint), C.POINTER(C.c_ulonglong)]
def read():
    torch.cuda.synchronize()
    bad, n = C.c_uint(0), C.c_ulonglong(0)
    assert L.mgu_diag_glds_read(C.byref(bad), C.byref(n)) == 0
    return bad.value, n.value
read()
cfg = (3, 2, 32, 4)
p = O.make_unet_params(*cfg, seed=3)
unet = mgunet.UNet(*cfg, compute_dtype=torch.bfloat16)
unet.load_state_dict(p)
unet = unet.to(cuda).eval()
tot = 0
for shape in ((8, 3, 512, 512), (2, 3, 200, 328), (1, 3, 1024, 1024)):
    x = torch.randn(shape, device=cuda)
    for _ in range(3):
        unet(x)
    bad, n = read()
    print(f"bf16 forward {shape}: {n} pieces compared, mismatch flag {bad}")
    tot += n
    assert bad == 0
assert tot > 0, "the LDS-DMA path did not run (is this the bf16 mode / the diag build?)"
print("LDS-DMA weight tiles: all landed tiles equal their global source")
