"""Worker of tests/test_gpu_parity.py::test_direct_rccl_weight_broadcast_two_ranks: one process per GPU, a raw RCCL communicator made with
ncclGetUniqueId / ncclCommInitRank (the id travels through a file), mcvd_model_broadcast_params from rank 0, one forward; rank r writes
its epsilon to <dir>/eps<r>.pt.  Usage: python tests/rccl_bcast_worker.py <rank> <world> <dir>"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    rank, world, d = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    torch.cuda.set_device(rank)
    from mcvd_pytorch_amd import _lib
    from mcvd_pytorch_amd.scorenet import HipScoreNet
    from oracle import synth
    rccl = C.CDLL("librccl.so.1")
    uid = (C.c_char * 128)()
    idf = os.path.join(d, "nccl_id.bin")
    if rank == 0:
        assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
        with open(idf + ".tmp", "wb") as f:
            f.write(bytes(uid))
        os.replace(idf + ".tmp", idf)
    else:
        t0 = time.time()
        while not os.path.exists(idf):
            assert time.time() - t0 < 120, "rank 0 never published the RCCL id"
            time.sleep(0.05)
        uid = (C.c_char * 128).from_buffer_copy(open(idf, "rb").read())

    class Uid(C.Structure):
        _fields_ = [("b", C.c_char * 128)]
    comm = C.c_void_p()
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, Uid, C.c_int]
    assert rccl.ncclCommInitRank(C.byref(comm), world, Uid(bytes(uid)), rank) == 0
    config = synth.make_config("tiny")
    config.device = f"cuda:{rank}"
    net = HipScoreNet(config)
    sd = synth.make_state_dict(config, seed=123 if rank == 0 else 999)          # the other ranks start from DIFFERENT weights
    net.load_state_dict(sd, strict=True)
    _lib.check(_lib.lib.mcvd_model_broadcast_params(net._model, comm, 0), "broadcast_params")
    _lib.check(_lib.lib.mcvd_model_finalize(net._model), "finalize")
    x, cond = synth.make_inputs(config, 2, seed=0)
    eps = net(x.cuda(), torch.tensor([990, 130]).cuda(), cond=cond.cuda())
    torch.cuda.synchronize()
    torch.save(eps.cpu(), os.path.join(d, f"eps{rank}.pt"))
    rccl.ncclCommDestroy(comm)


if __name__ == "__main__":
    main()
