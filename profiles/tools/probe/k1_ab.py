import glob, os, sys, subprocess
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for lib in ["amid_amd/libamid_hip.so"] + sorted(glob.glob(os.path.join(root, "profiles/tools/_diag/libamid_k1*.so"))):
    env = dict(os.environ, AMID_LIB_PATH=os.path.join(root, lib))
    print("==", os.path.basename(lib), flush=True)
    subprocess.run([sys.executable, os.path.join(root, "profiles/tools/k1_time.py")], env=env)
