import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from amid_amd._lib import lib
L = lib()
for four_min in (65536, 0):
  L.value("amid_sort_set_four_launch_min", four_min)
  print("four-launch sort for lists of >=", four_min, "indices")
  for n, n_rows, pad in ((26112, 894820, 0.89), (417792, 10_000_002, 0.0), (417792, 10_000_002, 0.89), (10752, 894820, 0.8)):
      g = torch.Generator().manual_seed(1)
      idx = torch.randint(0, n_rows, (n,), generator=g); idx[torch.rand(n, generator=g) < pad] = n_rows - 1
      idd = idx.int().cuda()
      ws = torch.zeros(L.value("amid_sort_unique_workspace_bytes", n), dtype=torch.uint8, device="cuda")
      o = [torch.zeros(n + 1, dtype=torch.int32, device="cuda") for _ in range(4)]; nu = torch.zeros(1, dtype=torch.int32, device="cuda")
      s = torch.cuda.current_stream().cuda_stream
      def run():
          L.call("amid_sort_unique_i32", idd.data_ptr(), n, n_rows, ws.data_ptr(), o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), o[3].data_ptr(), nu.data_ptr(), s)
      for _ in range(5): run()
      torch.cuda.synchronize()
      e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      e0.record()
      for _ in range(50): run()
      e1.record(); torch.cuda.synchronize()
      print(f"n={n} rows={n_rows} pad={pad}: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us per sort, U={int(nu)}")
