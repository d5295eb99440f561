"""Minimal ctypes binding of RCCL (librccl.so) for the token-parallel prefill's ONE collective: creating a communicator whose
handle goes to the C++ host loop (bitnet_host_prefill_sharded + bitnet_host_rccl_allgather) -- what a Rust host would do with
the same three calls.  The unique id travels over torch.distributed (any backend) because the ranks already rendezvous there.
Plumbing only."""
from __future__ import annotations

import ctypes as C


class UniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]  # ncclUniqueId: NCCL_UNIQUE_ID_BYTES = 128


def _lib():
    for name in ("librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"):
        try:
            return C.CDLL(name, mode=C.RTLD_GLOBAL)
        except OSError:
            continue
    raise OSError("librccl.so not found")


def version():
    """ncclGetVersion of the librccl.so this process would use, as "major.minor.patch" (None when the library is absent)."""
    try:
        L = _lib()
        v = C.c_int(0)
        L.ncclGetVersion.argtypes = [C.POINTER(C.c_int)]
        if L.ncclGetVersion(C.byref(v)) != 0:
            return None
    except (OSError, AttributeError):
        return None
    n = v.value
    return f"{n // 10000}.{n // 100 % 100}.{n % 100}" if n >= 10000 else f"{n // 1000}.{n // 100 % 10}.{n % 100}"


class Comm:
    def __init__(self, rank: int, world: int):
        import torch
        import torch.distributed as dist

        L = self.L = _lib()
        L.ncclGetUniqueId.argtypes = [C.POINTER(UniqueId)]
        L.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
        L.ncclCommDestroy.argtypes = [C.c_void_p]
        L.ncclGetErrorString.restype = C.c_char_p
        uid = UniqueId()
        if rank == 0:
            self._check(L.ncclGetUniqueId(C.byref(uid)))
        if world > 1:  # a single-rank communicator (tests) needs no exchange
            dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
            t = torch.tensor(list(bytes(uid)), dtype=torch.uint8, device=dev)
            dist.broadcast(t, 0)
            C.memmove(C.byref(uid), bytes(t.cpu().tolist()), 128)
        self.comm = C.c_void_p()
        self._check(L.ncclCommInitRank(C.byref(self.comm), world, uid, rank))

    def _check(self, rc: int) -> None:
        if rc != 0:
            raise RuntimeError(f"RCCL error {rc}: {self.L.ncclGetErrorString(rc).decode()}")

    @property
    def handle(self) -> int:
        return int(self.comm.value)

    def close(self) -> None:
        if self.comm:
            self.L.ncclCommDestroy(self.comm)
            self.comm = C.c_void_p()
