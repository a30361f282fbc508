"""A torch-free host (what a C caller looks like) driving the C-ABI's RCCL binding on one GPU: world of ONE rank,
device memory through the HIP runtime directly.  Prints 'comm ok' (or 'comm unsupported' when librccl.so is absent)."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(ROOT, "vittles_amd", "libvittles_hip.so"))
hip = ctypes.CDLL("libamdhip64.so")
vp, i64 = ctypes.c_void_p, ctypes.c_int64
hip.hipMalloc.argtypes = [ctypes.POINTER(vp), ctypes.c_size_t]
hip.hipMemcpy.argtypes = [vp, vp, ctypes.c_size_t, ctypes.c_int]
hip.hipFree.argtypes = [vp]
lib.vt_comm_unique_id.argtypes = [vp]
lib.vt_comm_init.argtypes = [ctypes.POINTER(vp), ctypes.c_int, ctypes.c_int, vp]
lib.vt_comm_destroy.argtypes = [vp]
lib.vt_allreduce_sum.argtypes = [vp, vp, i64, vp]


def ok(code, what):
    if code != 0:
        raise SystemExit("%s failed with %d" % (what, code))


ok(hip.hipSetDevice(0), "hipSetDevice")
uid = ctypes.create_string_buffer(128)
rc = lib.vt_comm_unique_id(uid)
if rc == -2:
    print("comm unsupported")
    sys.exit(0)
ok(rc, "vt_comm_unique_id")
comm = vp()
ok(lib.vt_comm_init(ctypes.byref(comm), 1, 0, uid), "vt_comm_init")
n = 100 * 100 + 100                     # [H | g] of the headline shape
host = np.random.default_rng(0).standard_normal(n)
dev = vp()
ok(hip.hipMalloc(ctypes.byref(dev), n * 8), "hipMalloc")
ok(hip.hipMemcpy(dev, host.ctypes.data_as(vp), n * 8, 1), "hipMemcpy H2D")
for _ in range(3):
    ok(lib.vt_allreduce_sum(comm, dev, n, None), "vt_allreduce_sum")
ok(hip.hipDeviceSynchronize(), "hipDeviceSynchronize")
back = np.empty(n)
ok(hip.hipMemcpy(back.ctypes.data_as(vp), dev, n * 8, 2), "hipMemcpy D2H")
assert np.array_equal(back, host), "a sum over one rank must return the buffer unchanged"
assert lib.vt_allreduce_sum(comm, None, 5, None) == -1 and lib.vt_allreduce_sum(None, dev, 5, None) == -1
ok(lib.vt_allreduce_sum(comm, dev, 0, None), "empty all-reduce")
ok(lib.vt_comm_destroy(comm), "vt_comm_destroy")
hip.hipFree(dev)
print("comm ok")
