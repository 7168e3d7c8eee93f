import gzip, json, os, pickle, shutil, sys, tempfile, time
root_repo = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, root_repo); sys.path.insert(0, os.path.join(root_repo, "tools")); sys.path.insert(0, os.path.join(root_repo, "tests"))
from make_synthetic_archive import make
from kurosiwo_amd.config import load_json5
TRAIN, VAL, TEST = [101, 102, 103, 104], [201], [301]
arch = tempfile.mkdtemp(prefix="ks_soak_")
os.makedirs(os.path.join(arch, "pickle"))
tr, _ = make(arch, TRAIN, tiles_per_act=64, seed=1)
te, _ = make(arch, VAL + TEST, tiles_per_act=24, seed=2)
pickle.dump(tr, gzip.open(os.path.join(arch, "pickle", "train.gz"), "wb"))
pickle.dump(te, gzip.open(os.path.join(arch, "pickle", "test.gz"), "wb"))
work = tempfile.mkdtemp(prefix="ks_run_")
shutil.copytree(os.path.join(root_repo, "configs"), os.path.join(work, "configs"))
dc = load_json5(os.path.join(work, "configs", "train", "data_config.json"))
dc.update(train_acts=TRAIN, val_acts=VAL, test_acts=TEST, train_pickle=os.path.join(arch, "pickle", "train.gz"), test_pickle=os.path.join(arch, "pickle", "test.gz"))
json.dump(dc, open(os.path.join(work, "configs", "train", "data_config.json"), "w"))
cc = load_json5(os.path.join(work, "configs", "config.json")); cc["root_path"] = arch
json.dump(cc, open(os.path.join(work, "configs", "config.json"), "w"))
tc = load_json5(os.path.join(work, "configs", "train", "train_config.json")); tc["epochs"] = 3
json.dump(tc, open(os.path.join(work, "configs", "train", "train_config.json"), "w"))
os.chdir(work); os.environ["KSMI_DATA"] = "archive"
import main as entry
t = time.time()
miou = entry.main(["--method", "snunet", "--inputs", "pre_event_1", "post_event", "--batch_size", "16", "--dem"])
print("SOAK ok: mIoU", miou, "wall", round(time.time() - t, 1), "s for 3 epochs of", len(tr), "cells")
