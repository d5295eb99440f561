"""CPU: the partition logic of the token-parallel prefill (bitnet-rs_amd/prefill_parallel.py) and
its one collective over gloo, world_size 2 (tests/_dist_worker.py)."""
import importlib

import numpy as np
import pytest

tp = importlib.import_module("bitnet-rs_amd.prefill_parallel")


@pytest.mark.parametrize("world,T", [(1, 128), (2, 256), (4, 4096), (8, 8192), (8, 1024)])
def test_zigzag_plan_covers_and_balances(world, T):
    plan = tp.zigzag_plan(T, world)
    pos = np.concatenate([tp.local_positions(p) for p in plan])
    assert sorted(pos.tolist()) == list(range(T))                       # every token exactly once
    assert len({len(tp.local_positions(p)) for p in plan}) == 1         # equal rows per rank (all_gather_into_tensor)
    assert all(s % 64 == 0 and n % 64 == 0 for p in plan for s, n in p)  # 64-token attention blocks
    # causal attention work per rank = sum over its queries of (position + 1): equal up to the chunk granularity
    work = [int((tp.local_positions(p) + 1).sum()) for p in plan]
    assert max(work) - min(work) <= 0.001 * max(work) + T
    assert plan[0][1][0] + plan[0][1][1] == T                            # the last token is rank 0's last local row
    # gathered -> absolute order
    gathered = [tp.local_positions(p).astype(np.float32)[:, None] for p in plan]
    out = np.full((T, 1), -1.0, np.float32)
    tp.scatter_gathered(gathered, plan, out)
    assert np.array_equal(out[:, 0], np.arange(T, dtype=np.float32))


def test_zigzag_plan_rejects_ragged_lengths():
    for world, T in ((2, 100), (8, 4096 + 64), (1, 0), (0, 128)):
        with pytest.raises(ValueError, match="multiple of"):
            tp.zigzag_plan(T, world)
