"""Static resource sheet of every HIP kernel for gfx950 (no GPU needed): VGPRs (architectural + accumulation), SGPRs, LDS, scratch and spills from
the code object metadata hipcc emits, and the occupancy they allow on a CDNA4 CU (512 VGPRs per SIMD lane in 8-register granules, at most 8
waves per SIMD; 160 KB LDS per CU).  Writes profiles/<name>.

    python tools/kernel_resources.py profiles/r03_kernel_resources_gfx950.txt
"""
import glob
import os
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "gs-sdf_amd", "csrc")
NO_CONTRACT = {"projection", "binning", "radix", "occupancy", "marching_cubes", "refine"}      # as the Makefile builds them
NO_SLP = {"raster_fwd", "raster_bwd"}
PAT = re.compile(r"- \.agpr_count:\s+(\d+).*?\.group_segment_fixed_size:\s+(\d+).*?\.max_flat_workgroup_size:\s+(\d+).*?\.name:\s+(\S+).*?"
                 r"\.private_segment_fixed_size:\s+(\d+).*?\.sgpr_count:\s+(\d+).*?\.sgpr_spill_count:\s+(\d+).*?\.vgpr_count:\s+(\d+).*?"
                 r"\.vgpr_spill_count:\s+(\d+)", re.S)


def assemble(src, out):
    stem = os.path.basename(src)[:-4]
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", f"-I{ROOT}/include", f"-I{CSRC}", "--cuda-device-only", "-S", src, "-o", out]
    if stem in NO_CONTRACT:
        cmd.insert(4, "-ffp-contract=off")
    if stem in NO_SLP:
        cmd.insert(4, "-fno-slp-vectorize")
    subprocess.run(cmd, check=True, capture_output=True, cwd=CSRC)
    return out


def main(path):
    rows = []
    with tempfile.TemporaryDirectory() as d:
        srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
        with ThreadPoolExecutor(8) as ex:
            outs = list(ex.map(lambda s: assemble(s, os.path.join(d, os.path.basename(s)[:-4] + ".s")), srcs))
        for o in outs:
            text = open(o).read()
            for m in PAT.finditer(text):
                ag, lds, wg, name, priv, sg, sgs, vg, vgs = m.groups()
                dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
                dem = re.sub(r"\(.*", "", dem).replace("void ", "").replace("gsdf::", "")
                vg, lds, wg = int(vg), int(lds), int(wg)
                waves_v = min(8, 512 // (-(-max(vg, 1) // 8) * 8))
                wg_waves = -(-wg // 64)
                wgs_lds = (160 * 1024) // lds if lds else 99
                waves_cu = min(waves_v * 4, wgs_lds * wg_waves, 32)
                rows.append((os.path.basename(o)[:-2], dem, vg, int(ag), int(sg), lds, int(priv), int(vgs), int(sgs), wg, waves_v, waves_cu))
    with open(path, "w") as f:
        f.write("# hipcc --offload-arch=gfx950 -O3, code-object metadata (tools/kernel_resources.py); vgpr = architectural + accumulation registers per lane;\n"
                "# waves/SIMD = what the VGPR count allows (<= 8); waves/CU = min of that x 4 SIMDs and the workgroups 160 KB of LDS hold (launch bounds as compiled)\n")
        f.write(f"{'file':16s} {'kernel':56s} {'vgpr':>5s} {'agpr':>5s} {'sgpr':>5s} {'lds B':>7s} {'scratch B':>9s} {'vgpr spills':>11s} {'wg':>5s} {'waves/SIMD':>10s} {'waves/CU':>8s}\n")
        for r in rows:
            f.write(f"{r[0]:16s} {r[1][:56]:56s} {r[2]:5d} {r[3]:5d} {r[4]:5d} {r[5]:7d} {r[6]:9d} {r[7]:11d} {r[9]:5d} {r[10]:10d} {r[11]:8d}\n")
    print(path, len(rows), "kernels;", sum(1 for r in rows if r[7]), "with VGPR spills")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "kernel_resources_gfx950.txt"))
