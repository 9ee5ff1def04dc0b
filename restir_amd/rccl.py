"""Plumbing around the strip driver's transports (include/restir_hip.h rs_comm_*): test / bench harness, not product.

* `RcclComm`: an ncclComm_t made the way a C++ caller of INTEGRATION.md makes it -- ncclGetUniqueId on rank 0, the 128-byte id handed
  to the other ranks by whatever side channel the job has (here: a callable, e.g. a torch.distributed object broadcast over gloo),
  ncclCommInitRank on every rank -- through ctypes on the copy of librccl the process already holds (PyTorch's wheel bundles one;
  a second copy from /opt/rocm/lib in the same process would not know the first one's communicators).
* `GlooTransport`: host callbacks for rs_comm_create that stage through host memory and torch.distributed point-to-point calls: the
  rehearsal of the multi-process path on a box whose ranks share one card (RCCL refuses two ranks on one device).
"""
import ctypes as C
import os


class NcclUniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]              # NCCL_UNIQUE_ID_BYTES


def loaded_librccl_path():
    """Path of the librccl this process has mapped (None if none)."""
    try:
        with open("/proc/self/maps") as fh:
            for line in fh:
                p = line.split()[-1]
                if "librccl" in os.path.basename(p):
                    return p
    except OSError:
        pass
    return None


class RcclComm:
    """ncclComm_t of `world` ranks.  broadcast(raw: bytes or None) -> bytes returns rank 0's bytes on every rank."""

    def __init__(self, rank, world, broadcast):
        path = loaded_librccl_path()
        if path is None:
            C.CDLL("librccl.so.1", mode=os.RTLD_NOW | os.RTLD_LOCAL)
            path = loaded_librccl_path() or "librccl.so.1"
        self.path = path
        self.lib = C.CDLL(path)
        self.lib.ncclGetUniqueId.argtypes = [C.POINTER(NcclUniqueId)]
        self.lib.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, NcclUniqueId, C.c_int]
        self.lib.ncclCommDestroy.argtypes = [C.c_void_p]
        self.lib.ncclGetErrorString.argtypes = [C.c_int]
        self.lib.ncclGetErrorString.restype = C.c_char_p
        self.rank, self.world = rank, world
        uid = NcclUniqueId()
        if rank == 0:
            self._check(self.lib.ncclGetUniqueId(C.byref(uid)), "ncclGetUniqueId")
        raw = broadcast(bytes(uid) if rank == 0 else None)
        assert len(raw) == C.sizeof(uid)
        C.memmove(C.byref(uid), raw, C.sizeof(uid))
        self.handle = C.c_void_p()
        self._check(self.lib.ncclCommInitRank(C.byref(self.handle), world, uid, rank), "ncclCommInitRank")

    def _check(self, code, what):
        if code != 0:
            raise RuntimeError("%s failed: %s" % (what, self.lib.ncclGetErrorString(code).decode()))

    def destroy(self):
        if self.handle:
            self.lib.ncclCommDestroy(self.handle)
            self.handle = C.c_void_p()


class GlooTransport:
    """send / recv / group_begin / group_end for capi.Comm(rank, world, ...): device buffer -> host -> torch.distributed (gloo)."""

    def __init__(self, capi, dist, torch):
        self.capi, self.dist, self.torch = capi, dist, torch
        self.ops, self.recvs, self.keep = [], [], []

    def send(self, ptr, nbytes, peer):
        t = self.torch.empty(nbytes, dtype=self.torch.uint8, device="cuda")
        self.capi.hip_memcpy_d2d(t.data_ptr(), ptr, nbytes)
        h = t.cpu(); self.keep.append(h)
        self.ops.append(self.dist.isend(h, peer))

    def recv(self, ptr, nbytes, peer):
        h = self.torch.empty(nbytes, dtype=self.torch.uint8)
        self.ops.append(self.dist.irecv(h, peer)); self.recvs.append((ptr, h))

    def begin(self):
        self.ops, self.recvs, self.keep = [], [], []

    def end(self):
        for w in self.ops:
            w.wait()
        for ptr, h in self.recvs:
            d = h.cuda()
            self.capi.hip_memcpy_d2d(ptr, d.data_ptr(), d.numel())
            self.torch.cuda.synchronize()

    def comm(self, rank, world):
        return self.capi.Comm(rank, world, self.send, self.recv, self.begin, self.end)
