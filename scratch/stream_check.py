"""A handle on a caller-provided (torch) stream, with the caller's kernels interleaved on that stream, matches the default stream."""
import sys; sys.path.insert(0, '/root/repo')
import numpy as np, torch
from real_robots_amd import _native as nat
from real_robots_amd.batched import BatchedREALRobotEnv
from real_robots_amd.distributed import synthetic_actions
N = 256
s = torch.cuda.Stream()
a = BatchedREALRobotEnv(N, objects=3, width=128, height=128)
b = BatchedREALRobotEnv(N, objects=3, width=128, height=128, stream=s.cuda_stream)
ids = range(N)
acc = torch.zeros(N, 128, 128, 3, device='cuda')
rgb_b = torch.as_tensor(b.device_buffer(nat.F_RGB), device='cuda')
for t in range(200):
    cmd = synthetic_actions(ids, t, seed=7) * 0.8
    a.step(cmd, render=True)
    with torch.cuda.stream(s):
        c = torch.from_numpy(cmd).cuda(non_blocking=False)
        b.step(device_ptr=c.data_ptr(), render=True)
        acc += rgb_b.float()                      # a consumer on the same stream, right behind the step
        s.synchronize() if t % 50 == 49 else None
s.synchronize(); torch.cuda.synchronize()
print("state equal", bool((a.state == b.state).all()), "images equal", bool((a.host(nat.F_RGB) == b.host(nat.F_RGB)).all()))
ref = torch.zeros_like(acc)
print("consumer saw complete frames (mean grey %.3f)" % float(acc.mean() / 200))
