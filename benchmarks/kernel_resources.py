"""Register / scratch / LDS use of every kernel of one translation unit, as the compiler
reports it for gfx950 (-Rpass-analysis=kernel-resource-usage):

    python benchmarks/kernel_resources.py openseize_amd/csrc/chain_spec.hip [filter]
"""
import re
import subprocess
import sys


def main():
    src = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-c", src,
           "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"] + sys.argv[3:]
    out = subprocess.run(cmd, capture_output=True, text=True).stderr
    rows, cur = [], None
    for line in out.splitlines():
        m = re.search(r"remark: +([A-Za-z ]+?)(?: \[bytes/lane\]| \[waves/SIMD\]| \[bytes/block\])?: (\S+)", line)
        if not m:
            if "error" in line:
                print(line)
            continue
        key, val = m.group(1).strip(), m.group(2)
        if key == "Function Name":
            cur = {"name": subprocess.run(["c++filt", val], capture_output=True, text=True).stdout.strip()}
            rows.append(cur)
        elif cur is not None:
            cur[key] = val
    print(f"{'kernel':70s} {'VGPR':>5s} {'AGPR':>5s} {'spill':>6s} {'scratch':>8s} {'SGPR':>5s} {'occ':>4s} {'LDS':>7s}")
    for r in rows:
        if flt and flt not in r["name"]:
            continue
        print(f"{r['name'][:70]:70s} {r.get('VGPRs', '?'):>5s} {r.get('AGPRs', '?'):>5s} "
              f"{r.get('VGPRs Spill', '?'):>6s} {r.get('ScratchSize', '?'):>8s} {r.get('SGPRs', '?'):>5s} "
              f"{r.get('Occupancy', '?'):>4s} {r.get('LDS Size', '?'):>7s}")


if __name__ == "__main__":
    main()
