"""Turns two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of `bench.py` into HBM bytes per launch
for every kernel family, following /opt/skills/guides/MI355X_MICROARCH.md section HBM:

* FETCH_SIZE / WRITE_SIZE are in kilobytes (x1024);
* on gfx950 FETCH_SIZE tallies the 128-byte requests of wide coalesced reads at 64 bytes, so the
  read side is DOUBLED; WRITE_SIZE is exact for 16-byte-per-lane streaming stores.

Kernel launches are mapped to the engine's kernel-family names by replaying the launch order of one
infer (the per-launch family list is written by bench.py --dump-launch-order).

usage: python tools/pmc_traffic.py <fetch_dir> <write_dir> <launch_order.json> <out.json> batch precision preset
"""
import collections
import csv
import glob
import json
import sys


def load(dirname, counter):
    f = glob.glob(f"{dirname}/**/*counter_collection.csv", recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    return rows


def main():
    fetch_dir, write_dir, order_path, out_path, batch, precision, preset = sys.argv[1:8]
    order = json.load(open(order_path))  # {"families": [...per launch of ONE infer...], "infers": n}
    fam = order["families"]
    res = collections.defaultdict(lambda: {"fetch_kb": 0.0, "write_kb": 0.0, "launches": 0})
    for dirname, counter, key in ((fetch_dir, "FETCH_SIZE", "fetch_kb"), (write_dir, "WRITE_SIZE", "write_kb")):
        # one dispatch per launch of a family: the assembly attention kernel (`md_attn577_bf16`) stands for its launch, the flag-consuming
        # `attention_redo_kernel` behind it (4 us, no traffic on trained weights) is left out
        rows = [r for r in load(dirname, counter) if r["Kernel_Name"].startswith(("void md::", "md::", "md_attn577"))
                and "attention_redo" not in r["Kernel_Name"]]  # (the scan and the recompute kernel)
        # keep only the launches of whole infers at the END of the run (the timed steps)
        n = len(fam) * order["infers"]
        rows = rows[-n:]
        assert len(rows) == n, (len(rows), n)
        for i, r in enumerate(rows):
            f = fam[i % len(fam)]
            res[f][key] += float(r["Counter_Value"])
            if key == "fetch_kb":
                res[f]["launches"] += 1
    out = {"batch": int(batch), "precision": precision, "preset": preset, "kernels": {},
           "note": "read bytes = 2 * FETCH_SIZE * 1024 (gfx950 correction), write bytes = WRITE_SIZE * 1024"}
    for f, v in res.items():
        rd, wr = 2.0 * v["fetch_kb"] * 1024, v["write_kb"] * 1024
        out["kernels"][f] = {"launches": v["launches"], "hbm_read_bytes_per_launch": rd / v["launches"],
                             "hbm_write_bytes_per_launch": wr / v["launches"],
                             "hbm_bytes_per_launch": (rd + wr) / v["launches"]}
    json.dump(out, open(out_path, "w"), indent=1)
    for f, v in sorted(out["kernels"].items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"]):
        print(f"{f:20s} launches {v['launches']:5d}  read {v['hbm_read_bytes_per_launch'] / 1e6:9.2f} MB  write {v['hbm_write_bytes_per_launch'] / 1e6:9.2f} MB")


if __name__ == "__main__":
    main()
