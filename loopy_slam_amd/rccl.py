"""RCCL collectives enqueued DIRECTLY on the launch stream (ctypes over librccl's C API: ncclCommInitRank / ncclAllReduce / ncclBroadcast).

torch.distributed's ProcessGroupNCCL runs every collective on a stream of its own and brackets it with two event hand-overs from and
to the caller's stream; on this chip a cross-stream event is a 6-8 us bubble on the stream that records or waits, and
the data-parallel mapping iteration has ONE small all-reduce (2-4 MB) on its critical path between the backward and the Adam step: measured
with one rank (nothing on the wire) the exchange machinery cost 23 us per iteration that way.  Here the collective is one more launch in
the stream's own order - no second stream, no events - on a communicator of this library's own; the unique id travels through the
torch.distributed group the process already has (any backend).

Only what the hot path exchanges: float32 SUM and uint8 MAX all-reduce (in place), broadcast from a root."""
import ctypes as C
import os

import torch
import torch.distributed as dist

NCCL_UINT8, NCCL_FLOAT32 = 1, 7
NCCL_SUM, NCCL_MAX = 0, 2


class _UniqueId(C.Structure):
    _fields_ = [('internal', C.c_byte * 128)]


def _load():
    # the copy torch itself links (its lib directory) first: one RCCL per process
    cands = [os.path.join(os.path.dirname(torch.__file__), 'lib', 'librccl.so'), 'librccl.so.1', 'librccl.so']
    last = None
    for c in cands:
        try:
            return C.CDLL(c)
        except OSError as e:
            last = e
    raise OSError(f'librccl not found: {last}')


class RcclComm:
    def __init__(self, rank, world, device):
        self.rank, self.world, self.device = rank, world, torch.device(device)
        lib = self.lib = _load()
        lib.ncclGetErrorString.restype = C.c_char_p
        lib.ncclGetUniqueId.argtypes = [C.POINTER(_UniqueId)]
        lib.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _UniqueId, C.c_int]
        lib.ncclAllReduce.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        lib.ncclBroadcast.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        lib.ncclCommDestroy.argtypes = [C.c_void_p]
        uid = _UniqueId()
        if rank == 0:
            self._check(lib.ncclGetUniqueId(C.byref(uid)), 'ncclGetUniqueId')
        raw = torch.tensor(list(bytes(uid)), dtype=torch.uint8)
        if world > 1:
            if dist.get_backend() == 'nccl':
                raw = raw.to(self.device)
            dist.broadcast(raw, src=0)
        C.memmove(C.byref(uid), bytes(raw.cpu().tolist()), 128)
        self.comm = C.c_void_p()
        with torch.cuda.device(self.device):
            self._check(lib.ncclCommInitRank(C.byref(self.comm), world, uid, rank), 'ncclCommInitRank')

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError(f'{what} failed: {self.lib.ncclGetErrorString(rc).decode()}')

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def all_reduce(self, t, op='sum'):
        """In place on the CURRENT stream.  t: contiguous float32 (sum) or uint8 (max) device tensor."""
        assert t.is_cuda and t.is_contiguous()
        dt = {torch.float32: NCCL_FLOAT32, torch.uint8: NCCL_UINT8}[t.dtype]
        self._check(self.lib.ncclAllReduce(t.data_ptr(), t.data_ptr(), t.numel(), dt, NCCL_SUM if op == 'sum' else NCCL_MAX, self.comm, self._stream()),
                    'ncclAllReduce')
        return t

    def broadcast(self, t, src=0):
        assert t.is_cuda and t.is_contiguous()
        dt = {torch.float32: NCCL_FLOAT32, torch.uint8: NCCL_UINT8}[t.dtype]
        self._check(self.lib.ncclBroadcast(t.data_ptr(), t.data_ptr(), t.numel(), dt, src, self.comm, self._stream()), 'ncclBroadcast')
        return t

    def close(self):
        if self.comm:
            torch.cuda.synchronize(self.device)
            self.lib.ncclCommDestroy(self.comm)
            self.comm = C.c_void_p()
