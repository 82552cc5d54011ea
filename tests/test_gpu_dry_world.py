"""bench.py --dry-world N: the sharding arithmetic of an N-rank run (map shard bounds with a remainder, base indices,
per-rank seeds), executed rank by rank on one GPU; the merged partial results must equal the unsharded search."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world", [2, 5])
def test_dry_world_merges_to_the_unsharded_result(world):
    run = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-world", str(world)], capture_output=True,
                         text=True, timeout=170, cwd=ROOT)
    assert run.returncode == 0, run.stdout + run.stderr
    d = json.loads(run.stdout.strip().splitlines()[-1])["dry_world"]
    assert d["world"] == world and d["merged_equals_unsharded"] and d["frame_seeds_distinct"]
    bounds = d["shards"]
    assert bounds[0][0] == 0 and sum(n for _, n in bounds) == d["map_rows"]
    assert all(bounds[i][0] + bounds[i][1] == bounds[i + 1][0] for i in range(world - 1))
