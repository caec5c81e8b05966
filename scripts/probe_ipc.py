"""Feasibility probe: fine-grained allocation + HIP IPC handles between two processes (gloo for the handshake)."""
import ctypes as C, os, sys, time
import torch, torch.distributed as dist
hip = C.CDLL("libamdhip64.so")
class Handle(C.Structure):
    _fields_ = [("r", C.c_ubyte * 64)]
hip.hipIpcOpenMemHandle.argtypes = [C.POINTER(C.c_void_p), Handle, C.c_uint]
hip.hipIpcGetMemHandle.argtypes = [C.POINTER(Handle), C.c_void_p]
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
torch.cuda.set_device(0)
torch.zeros(1, device="cuda")
def chk(rc, what):
    if rc != 0: raise RuntimeError(f"{what} -> {rc}")
hipDeviceMallocFinegrained = 0x1
p = C.c_void_p()
chk(hip.hipExtMallocWithFlags(C.byref(p), C.c_size_t(4096), C.c_uint(hipDeviceMallocFinegrained)), "extmalloc fine")
chk(hip.hipMemset(p, 0, C.c_size_t(4096)), "memset")
h = Handle()
chk(hip.hipIpcGetMemHandle(C.byref(h), p), "ipc get (fine-grained)")
d = C.c_void_p()
chk(hip.hipMalloc(C.byref(d), C.c_size_t(1 << 20)), "malloc")
hd = Handle()
chk(hip.hipIpcGetMemHandle(C.byref(hd), d), "ipc get (coarse)")
objs = [None] * world
dist.all_gather_object(objs, (bytes(h.r), bytes(hd.r)))
peer = (rank + 1) % world
ph = Handle(); C.memmove(ph.r, objs[peer][0], 64); pd = Handle(); C.memmove(pd.r, objs[peer][1], 64)
rp, rd = C.c_void_p(), C.c_void_p()
chk(hip.hipIpcOpenMemHandle(C.byref(rp), ph, 1), "ipc open fine")
chk(hip.hipIpcOpenMemHandle(C.byref(rd), pd, 1), "ipc open coarse")
# write a value into the peer's buffers, then read ours back after a barrier
val = (C.c_int * 1)(1000 + rank)
chk(hip.hipMemcpy(rp, val, C.c_size_t(4), C.c_int(1)), "h2d remote fine")
chk(hip.hipMemcpy(rd, val, C.c_size_t(4), C.c_int(1)), "h2d remote coarse")
chk(hip.hipDeviceSynchronize(), "sync")
dist.barrier()
out = (C.c_int * 1)(); out2 = (C.c_int * 1)()
chk(hip.hipMemcpy(out, p, C.c_size_t(4), C.c_int(2)), "d2h")
chk(hip.hipMemcpy(out2, d, C.c_size_t(4), C.c_int(2)), "d2h")
print(f"rank {rank}: fine-grained flag got {out[0]}, coarse buffer got {out2[0]} (expected {1000 + (rank - 1) % world})", flush=True)
dist.barrier()
chk(hip.hipIpcCloseMemHandle(rp), "close"); chk(hip.hipIpcCloseMemHandle(rd), "close")
dist.destroy_process_group()
