import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import torch
import toyfhe_jl_amd as tf
from toyfhe_jl_amd import dist as tdist
print("torch rccl:", os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"), os.environ.get("TFHE_RCCL_LIB"))
torch.cuda.init()
ctx = tf.Context(64, [tf.nextprime(2**40 + 1, 1, 128)])
comm = tdist.make_comm()
a = np.arange(1000, dtype=np.uint64)
src, dst = tf.DeviceBuffer.from_numpy(a), tf.DeviceBuffer(1000)
comm.gather(ctx, src.ptr, dst.ptr, 1000); ctx.sync()
print("OK", np.array_equal(dst.to_numpy(), a))
