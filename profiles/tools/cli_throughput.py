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


if __name__ == "__main__":
    main_dr() if "--dr" in sys.argv else main()
