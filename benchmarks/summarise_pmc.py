"""Reduce the rocprofv3 --pmc passes of benchmarks/collect_pmc.sh to one table
per kernel under profiles/:

    python benchmarks/summarise_pmc.py gpurun_out/pmc rNN

For every kernel the LARGEST launches are kept (grid >= 1/2 of the kernel's
largest grid: warm-up and remainder launches of the same kernel are dropped)
and every counter is averaged over them.  Also written per kernel: VGPRs, LDS
per workgroup, mean duration of the profiled launches, and derived ratios
(SQ units are quad-cycles: MI355X_MICROARCH.md, cycle-constants table):

  busy      = SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES   issue-active share of wave time
  valu      = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES
  wait_any  = SQ_WAIT_ANY / SQ_WAVE_CYCLES          waves parked (s_waitcnt, barrier)
  lds_conf  = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
  lds_array = SQ_LDS_IDX_ACTIVE / 256 CUs / (SQ_BUSY_CYCLES / 32 shader engines): the share of the
              launch a CU's LDS array is busy (specmix before its 16-byte accesses: 0.78; this, not
              SQ_ACTIVE_INST_LDS / SQ_WAVE_CYCLES, says whether a kernel is LDS-bound)
  hbm_bytes = (2 FETCH_SIZE + WRITE_SIZE) * 1024    (gfx950 FETCH_SIZE half count)
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def short(name):
    name = name.replace("osz::", "")
    cut = name.find("(")
    return name[:cut] if cut > 0 else name


def main():
    root, tag = sys.argv[1], sys.argv[2]
    out_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles")
    for group in ("bench", "sec"):
        rows = defaultdict(lambda: defaultdict(list))    # kernel -> counter -> [(grid, value, dur)]
        meta = {}
        # the newest file of every pass: gpurun merges a run's outputs into gpurun_out/ beside
        # those of earlier runs
        newest = {}
        for path in glob.glob(os.path.join(root, f"{group}_*", "**", "*counter_collection.csv"),
                              recursive=True):
            key = os.path.relpath(path, root).split(os.sep)[0]
            if key not in newest or os.path.getmtime(path) > os.path.getmtime(newest[key]):
                newest[key] = path
        for path in newest.values():
            with open(path, newline="") as fh:
                for r in csv.DictReader(fh):
                    k = short(r["Kernel_Name"])
                    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
                    rows[k][r["Counter_Name"]].append((int(r["Grid_Size"]), float(r["Counter_Value"]), dur))
                    meta[k] = (r["VGPR_Count"], r["Accum_VGPR_Count"], r["SGPR_Count"],
                               r["LDS_Block_Size"], r["Workgroup_Size"])
        if not rows:
            continue
        table = {}
        for k, ctrs in rows.items():
            big = max(g for vals in ctrs.values() for g, _, _ in vals)
            rec = {"vgpr": int(meta[k][0]), "agpr": int(meta[k][1]), "sgpr": int(meta[k][2]),
                   "lds_bytes": int(meta[k][3]), "wg_size": int(meta[k][4]), "grid": big}
            durs = []
            for c, vals in ctrs.items():
                keep = [(v, d) for g, v, d in vals if 2 * g >= big]
                rec[c] = sum(v for v, _ in keep) / len(keep)
                rec["launches_" + c] = len(keep)
                durs += [d for _, d in keep]
            rec["profiled_ms"] = sum(durs) / len(durs)
            wc = rec.get("SQ_WAVE_CYCLES")
            if wc:
                for nm, c in (("busy", "SQ_ACTIVE_INST_ANY"), ("valu", "SQ_ACTIVE_INST_VALU"),
                              ("wait_any", "SQ_WAIT_ANY"), ("wait_inst", "SQ_WAIT_INST_ANY"),
                              ("lds_active", "SQ_ACTIVE_INST_LDS"), ("vmem_active", "SQ_ACTIVE_INST_VMEM")):
                    if c in rec:
                        rec[nm] = rec[c] / wc
            if rec.get("SQ_LDS_IDX_ACTIVE"):
                rec["lds_conf"] = rec.get("SQ_LDS_BANK_CONFLICT", 0.0) / rec["SQ_LDS_IDX_ACTIVE"]
                if rec.get("SQ_BUSY_CYCLES"):
                    rec["lds_array"] = rec["SQ_LDS_IDX_ACTIVE"] / 256.0 / (rec["SQ_BUSY_CYCLES"] / 32.0)
            if "FETCH_SIZE" in rec and "WRITE_SIZE" in rec:
                rec["hbm_bytes"] = (2 * rec["FETCH_SIZE"] + rec["WRITE_SIZE"]) * 1024
            table[k] = {key: val for key, val in rec.items() if not key.startswith("launches_")}
        # keep the kernels that matter (>= 20 us)
        table = {k: v for k, v in table.items() if v["profiled_ms"] >= 0.02}
        dst = os.path.join(out_dir, f"{tag}_pmc_{group}.json")
        with open(dst, "w") as fh:
            json.dump(table, fh, indent=1, sort_keys=True)
        print(dst)
        for k, v in sorted(table.items(), key=lambda kv: -kv[1]["profiled_ms"]):
            print(f"{k[:60]:60s} {v['profiled_ms']:8.3f} ms vgpr {v['vgpr']:3d} lds {v['lds_bytes']:6d} "
                  f"busy {v.get('busy', 0):.2f} valu {v.get('valu', 0):.2f} wait {v.get('wait_any', 0):.2f} "
                  f"wait_inst {v.get('wait_inst', 0):.2f} lds {v.get('lds_active', 0):.2f} vmem {v.get('vmem_active', 0):.2f} "
                  f"conf {v.get('lds_conf', 0):.3f} lds_array {v.get('lds_array', 0):.2f} hbm {v.get('hbm_bytes', 0) / 1e9:.2f} GB")


if __name__ == "__main__":
    main()
