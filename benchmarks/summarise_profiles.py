"""Turn the rocprofv3 outputs of benchmarks/refresh_profiles.sh into the files
kept under profiles/: the per-kernel duration summary, the mean FETCH_SIZE /
WRITE_SIZE per kernel (separate --pmc passes) and traffic.json, the HBM bytes
per launch that bench.py reports as roofline.traffic.

    python benchmarks/summarise_profiles.py gpurun_out/prof rNN

HBM bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE under-counts
streaming reads by 1/2 on gfx950 (MI355X_MICROARCH.md, HBM section); the factor
is checked in the same run on checksum_kernel (known 2 GiB read) and
synth_normal_kernel (known 2 GiB write).
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def find(root, pattern):
    """Newest match: gpurun merges into gpurun_out/ without deleting earlier runs."""
    hits = glob.glob(os.path.join(root, "**", pattern), recursive=True)
    return max(hits, key=os.path.getmtime) if hits else None


def counter_means(path, counter):
    acc = defaultdict(lambda: [0, 0.0])
    if not path:
        return acc
    with open(path, newline="") as fh:
        for row in csv.DictReader(fh):
            if row.get("Counter_Name") != counter:
                continue
            a = acc[row["Kernel_Name"]]
            a[0] += 1
            a[1] += float(row["Counter_Value"])
    return acc


def kernel_sources_sha():
    """Fingerprint of the kernel sources the counters were collected with: bench.py
    reports the counters as stale (traffic = null) when the sources have changed."""
    import hashlib
    csrc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "openseize_amd", "csrc")
    h = hashlib.sha256()
    for name in sorted(os.listdir(csrc)):
        if name.endswith((".hip", ".h")):
            h.update(name.encode())
            h.update(open(os.path.join(csrc, name), "rb").read())
    return h.hexdigest()


def main():
    if sys.argv[1:] == ["--stamp"]:
        # the sources are what the committed counters were collected with: stamp them
        tpath = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", "traffic.json")
        traffic = json.load(open(tpath))
        traffic["_kernel_sources_sha256"] = kernel_sources_sha()
        json.dump(traffic, open(tpath, "w"), indent=1)
        return
    root, tag = sys.argv[1], sys.argv[2]
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles")
    stats = find(os.path.join(root, "stats"), "*kernel_stats.csv")
    if stats:
        with open(stats) as src, open(os.path.join(out, f"{tag}_kernel_stats_bench.csv"), "w") as dst:
            dst.write(src.read())
    fetch = counter_means(find(os.path.join(root, "fetch"), "*counter_collection.csv"), "FETCH_SIZE")
    write = counter_means(find(os.path.join(root, "write"), "*counter_collection.csv"), "WRITE_SIZE")
    # keys measured by an earlier run of another step variant (bench.py --unfused) stay
    tpath = os.path.join(out, "traffic.json")
    traffic = json.load(open(tpath)) if os.path.exists(tpath) else {}
    with open(os.path.join(out, f"{tag}_pmc_hbm_traffic.csv"), "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["kernel", "launches_fetch", "FETCH_SIZE_KB_mean", "launches_write",
                    "WRITE_SIZE_KB_mean", "hbm_bytes_per_launch(2*FETCH+WRITE)*1024"])
        for name in fetch:
            nf, sf = fetch[name]
            nw, sw = write.get(name, [0, 0.0])
            f, wr = sf / max(nf, 1), sw / max(nw, 1)
            hbm = (2 * f + wr) * 1024
            w.writerow([name, nf, round(f, 1), nw, round(wr, 1), int(hbm)])
            # the big launches only (the same kernels run tiny warm-up launches)
            if "sos_dual" in name and hbm > 1e9:
                traffic["sos_dual"] = hbm
            elif ("fir_oa_kernel" in name or "fir_nega_kernel" in name) and hbm > 1e9:
                traffic["fir_oa"] = hbm
            elif "sos_kernel" in name and "false, false" in name and hbm > 1e9:
                traffic["sos_fwd"] = hbm
            elif ("chain_kernel" in name or "chain_spec_kernel" in name) and hbm > 1e9:
                traffic["chain_fwd"] = hbm
            elif ("chain_zp_kernel" in name or "chain_zpn_kernel" in name) and hbm > 1e9:
                traffic["chain_zp"] = hbm
            elif "sos_split2_kernel<32, 4, true" in name and hbm > 1e9:
                traffic["sos_bwd_split"] = hbm
    if "chain_fwd" in traffic and "sos_bwd_split" in traffic:
        # one osz_chain_step = the fused forward kernel and the backward pass, side by side
        traffic["chain_step"] = traffic["chain_fwd"] + traffic["sos_bwd_split"]
    traffic["_note"] = ("HBM bytes per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
                        "(separate passes), (2*FETCH_SIZE + WRITE_SIZE)*1024; see "
                        f"{tag}_pmc_hbm_traffic.csv and profiles/README.md for the calibration.")
    # the secondary workloads' kernels (bench.py --workload welch / stft): bytes per launch
    # from the counter passes of benchmarks/collect_pmc.sh secondary, when reduced already
    sec = os.path.join(out, f"{tag}_pmc_sec.json")
    if os.path.exists(sec):
        pmc = json.load(open(sec))
        for key, name in (("spec_fused", "void spec_cube_kernel<0, false, true>"),
                          ("poly_block", "void poly_block_kernel<true, 256, 1>")):
            if name in pmc and pmc[name].get("hbm_bytes"):
                traffic[key] = pmc[name]["hbm_bytes"]
    traffic["_kernel_sources_sha256"] = kernel_sources_sha()
    with open(os.path.join(out, "traffic.json"), "w") as fh:
        json.dump(traffic, fh, indent=1)
    print(json.dumps(traffic, indent=1))


if __name__ == "__main__":
    main()
