"""Per-step kernel timeline from a rocprofv3 rocpd database (rocprofv3 --kernel-trace -d DIR -o NAME -- python3 bench.py ...):
prints, for one replayed step in the middle of the run (of the eight around the middle the shortest: under the profiler the last step of
a four-step graph carries the gap to the next graph launch), every kernel's start (us, relative to the step's first kernel), duration and
stream, so that gaps and overlaps between the main branch and the side-stream sort can be read off.
    python profiles/tools/step_timeline.py gpurun_out/tl/x_results.db [anchor-kernel-substring]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end, stream_id, grid_x * grid_y * grid_z, workgroup_x * workgroup_y * workgroup_z, lds_size from kernels order by start").fetchall()
# the step's first kernel: the step head of the folded step (round 5), else the packing launch
anchor = sys.argv[2] if len(sys.argv) > 2 else ("step_head" if any("step_head" in r[0] for r in rows) else "pack_indices")
starts = [i for i, r in enumerate(rows) if anchor in r[0]]
if len(starts) < 4:
    raise SystemExit(f"anchor {anchor!r} found {len(starts)} times")
mid = len(starts) // 2
cand = [(rows[starts[k + 1]][1] - rows[starts[k]][1], k) for k in range(max(0, mid - 4), min(len(starts) - 1, mid + 4))]
k = min(cand)[1]
i0, i1 = starts[k], starts[k + 1]
t0 = rows[i0][1]
print(f"step of {(rows[i1][1] - t0) / 1e3:.1f} us, {i1 - i0} kernels")
for name, s, e, st, gx, wx, lds in rows[i0:i1]:
    print(f"{(s - t0) / 1e3:8.1f} {(e - s) / 1e3:7.1f}  s{st}  {gx // max(wx, 1):5d} wg  lds {lds:6d}  {name[:70]}")
