"""End-to-end throughput of the train_sr.py CLI (loader included) on a synthetic CSV shaped like cloth_sport_train75
(15 403 rows, mean sequence length ~5.5, ids <= 42 441): `python profiles/tools/cli_throughput.py [extra CLI args]`."""
import os, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))


def write_csv(path, n, rng, ob_label=False, max_len=50):
    rows = ["user_id,seq_d1,seq_d2,domain_id" + (",ob_label" if ob_label else "")]
    for u in range(n):
        dom = int(rng.random() < 0.5)
        l1 = int(min(max_len, rng.poisson(4.5) + (2 if dom == 0 else 0)))
        l2 = int(min(max_len, rng.poisson(4.5) + (2 if dom == 1 else 0)))
        s1 = [int(x) for x in rng.integers(1, 21000, l1)]
        s2 = [int(x) for x in rng.integers(21000, 42441, l2)]
        rows.append(f'{u},"{s1}","{s2}",{dom}' + (f",{int(rng.random() < 0.6)}" if ob_label else ""))
    open(path, "w").write("\n".join(rows) + "\n")


def main_dr():
    """run.sh's command line (train_sr_dr.py, mybank, isItC, doubly-robust heads, 999 negatives at evaluation) on synthetic CSVs of
    the mybank size (60 k train rows each, seq_len 20)."""
    from amid_amd.train_sr_dr import main as cli
    rng = np.random.default_rng(0)
    tmp = tempfile.mkdtemp()
    root = os.path.join(tmp, "mybank_dataset")
    os.makedirs(root)
    write_csv(os.path.join(root, "toy_train25.csv"), 60600, rng, max_len=20)
    write_csv(os.path.join(root, "toy_train25_DR.csv"), 60600, rng, ob_label=True, max_len=20)
    write_csv(os.path.join(root, "toy_test.csv"), 4096, rng, max_len=20)
    t0 = time.perf_counter()
    cli(["--data_root", tmp, "-ds", "mybank", "-dm", "toy", "--overlap_ratio", "0.25", "--model", "sasrec", "--overlap", "True", "--isItC", "True",
         "--ts2", "0.4", "--neg_nums", "999", "--lr2", "0.01", "--dr_e_w", "0.01", "--bs", "256", "--seq_len", "20", "--emb_dim", "128",
         "--hid_dim", "32", "--epoch", "3", "--seeds", "1", "-md", os.path.join(tmp, "model")] + [a for a in sys.argv[1:] if a != "--dr"])
    print("wall", time.perf_counter() - t0)


def main():
    from amid_amd.train_sr import main as cli
    rng = np.random.default_rng(0)
    tmp = tempfile.mkdtemp()
    root = os.path.join(tmp, "amazon_dataset")
    os.makedirs(root)
    write_csv(os.path.join(root, "toy_train75.csv"), 15403, rng)
    write_csv(os.path.join(root, "toy_test.csv"), 2048, rng)
    t0 = time.perf_counter()
    cli(["--data_root", tmp, "-ds", "amazon", "-dm", "toy", "--overlap_ratio", "0.75", "--model", "sasrec", "--bs", "256", "--seq_len", "50",
         "--emb_dim", "128", "--hid_dim", "32", "--epoch", "4", "--neg_nums", "199", "--seeds", "1", "-md", os.path.join(tmp, "model")] + sys.argv[1:])
    print("wall", time.perf_counter() - t0)


def main_cfg1():
    """BASELINE.json configs[0] through the CLI, whole epochs with their evaluation (train_sr.py:130-351): a synthetic CSV pair of
    cloth_sport_train25's size -- 8 119 train rows = 126 batches of 64, 4 750 test rows = 74 batches, seq_len 50, emb_dim 64, 199 negatives
    at evaluation -- beside SURVEY.md section 6's figure for the reference on 8 CPU threads (72.2 s per epoch incl. eval).  Prints every
    epoch's train / eval seconds as the CLI logs them and their sum (the first epoch also captures the graphs)."""
    import logging, re
    from amid_amd.train_sr import main as cli
    rng = np.random.default_rng(0)
    tmp = tempfile.mkdtemp()
    root = os.path.join(tmp, "amazon_dataset")
    os.makedirs(root)
    write_csv(os.path.join(root, "toy_train25.csv"), 8119, rng)
    write_csv(os.path.join(root, "toy_test.csv"), 4750, rng)
    seen = []

    class Grab(logging.Handler):
        def emit(self, rec):
            seen.append(rec.getMessage())
    logging.getLogger().addHandler(Grab())
    t0 = time.perf_counter()
    cli(["--data_root", tmp, "-ds", "amazon", "-dm", "toy", "--overlap_ratio", "0.25", "--model", "sasrec", "--bs", "64", "--seq_len", "50",
         "--emb_dim", "64", "--hid_dim", "32", "--epoch", "4", "--neg_nums", "199", "--seeds", "1", "-md", os.path.join(tmp, "model")]
        + [a for a in sys.argv[1:] if a != "--cfg1"])
    wall = time.perf_counter() - t0
    tr = [float(m.group(1)) for s_ in seen for m in [re.search(r"epoch \d+: \d+ samples in ([0-9.]+) s", s_)] if m]
    ev = [float(m.group(1)) for s_ in seen for m in [re.search(r"evaluated \d+ samples x \d+ candidates in ([0-9.]+) s", s_)] if m]
    for k, (a, b) in enumerate(zip(tr, ev)):
        print(f"cfg 1 epoch {k}: train {a:.3f} s (126 batches of 64) + eval {b:.3f} s (74 batches x 200 candidates) = {a + b:.3f} s"
              + ("   [captures the graphs]" if k == 0 else ""))
    if len(tr) > 1:
        best = min(a + b for a, b in list(zip(tr, ev))[1:])
        print(f"cfg 1 epoch incl. eval, steady: {best:.3f} s   (reference on 8 CPU threads, SURVEY.md section 6: 72.2 s  ->  {72.2 / best:.0f} x)")
    print("wall (4 epochs, CSV parsing and set-up included)", round(wall, 2))


if __name__ == "__main__":
    main_cfg1() if "--cfg1" in sys.argv else main_dr() if "--dr" in sys.argv else main()
