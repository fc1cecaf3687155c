"""Which host mappings can hipHostRegister pin on this box?  (decides between zero-copy DMA out of
the shared-memory mailbox / ingestion ring and the pinned staging fallback)"""
import ctypes
import mmap

import numpy as np
import torch

hip = ctypes.CDLL("libamdhip64.so")
hip.hipHostRegister.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint]
hip.hipHostUnregister.argtypes = [ctypes.c_void_p]
hip.hipGetErrorString.restype = ctypes.c_char_p
hip.hipGetErrorString.argtypes = [ctypes.c_int]
torch.cuda.init()
torch.zeros(1, device="cuda")


def attempt(name, ptr, nbytes):
    for flags in (0, 1, 2, 3):
        rc = hip.hipHostRegister(ptr, nbytes, flags)
        print("%-28s flags=%d -> %d %s" % (name, flags, rc, hip.hipGetErrorString(rc).decode()), flush=True)
        if rc == 0:
            hip.hipHostUnregister(ptr)
            hip.hipGetLastError()
            break
        hip.hipGetLastError()


n = 8 << 20
a = np.zeros(n, dtype=np.uint8)
attempt("numpy heap", a.ctypes.data, n)
t = torch.zeros(n, dtype=torch.uint8).share_memory_()
print("shared tensor ptr %x (page aligned: %s)" % (t.data_ptr(), t.data_ptr() % 4096 == 0))
attempt("torch share_memory_", t.data_ptr(), n)
attempt("torch share_memory_ +4096", t.data_ptr() + 4096, n - 8192)
m = mmap.mmap(-1, n)
buf = (ctypes.c_char * n).from_buffer(m)
attempt("anonymous shared mmap", ctypes.addressof(buf), n)
import os
fd = os.open("/dev/shm/pda_probe", os.O_CREAT | os.O_RDWR)
os.ftruncate(fd, n)
m2 = mmap.mmap(fd, n)
buf2 = (ctypes.c_char * n).from_buffer(m2)
buf2[0] = b"x"
attempt("/dev/shm file mmap", ctypes.addressof(buf2), n)
os.unlink("/dev/shm/pda_probe")
print("sharing strategy:", torch.multiprocessing.get_sharing_strategy())
