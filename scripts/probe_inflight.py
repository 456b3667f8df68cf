"""What is the vector L1's outstanding-read capacity?  Three access patterns at full occupancy, to be run under
`rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum` and
`--pmc GRBM_GUI_ACTIVE`: reads in flight per CU = LATENCY_sum / (GRBM_GUI_ACTIVE / 8) / 256.
  1. read-only stream, 1 GiB (HBM latency), 32 / 16 / 8 resident waves per CU
  2. hashed 4-byte loads from a 4 MB table without any stream (L2-hit latency), 2048 workgroups
  3. the gather probe out[i] = table[idx[i]] (index + output streams + L2 gathers)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from loops_amd import probes as PR

sink = torch.zeros(16, device="cuda")
src = torch.empty(1 << 28, dtype=torch.float32, device="cuda").normal_()
for waves in (32, 16, 8):
    for _ in range(5):
        PR.stream_read_prefetch(src, sink, 0, 32, waves)
del src
table = torch.empty(1 << 20, dtype=torch.float32, device="cuda").normal_()
out = torch.empty(2048 * 256, dtype=torch.float32, device="cuda")
for blocks in (2048, 256):
    for _ in range(5):
        PR.address_rate(table, 2000, 1, blocks, out)
idx = torch.randint(0, 1 << 20, (1 << 24,), dtype=torch.int32, device="cuda")
g = torch.empty(1 << 24, dtype=torch.float32, device="cuda")
for _ in range(5):
    PR.gather(table, idx, g)
torch.cuda.synchronize()
