"""Register / LDS / scratch table of every kernel in libucd_hip (compile-time: no GPU needed).

Re-compiles each csrc/*.hip with ``-Rpass-analysis=kernel-resource-usage`` (objects discarded) and prints one line per
kernel: VGPRs, AGPRs, SGPRs, scratch bytes per lane (a non-zero value = spilling), occupancy in waves per SIMD and static LDS.
usage: python tools/kernel_resources.py > profiles/rNN_kernel_resources.txt"""
import glob
import os
import re
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "..", "ucd_amd", "csrc")
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "--cuda-device-only",
         "-Rpass-analysis=kernel-resource-usage", "-c", "-o", "/dev/null"]


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout
    return out.splitlines()


def main():
    print("# hipcc " + " ".join(FLAGS[:-3]) + "  (gfx950; 512 VGPRs + AGPRs per SIMD lane: occupancy 1 above 256 combined)")
    print("%-5s %-5s %-5s %-8s %-4s %-8s  %s" % ("VGPR", "AGPR", "SGPR", "scratch", "occ", "LDS", "kernel"))
    for path in sorted(glob.glob(os.path.join(SRC, "*.hip"))):
        err = subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, path], capture_output=True, text=True, cwd=SRC).stderr
        rows, cur = [], None
        for line in err.splitlines():
            m = re.search(r"remark:\s+(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|"
                          r"LDS Size \[bytes/block\]): (\S+)", line)
            if not m:
                continue
            if m.group(1) == "Function Name":
                cur = {"name": m.group(2)}
                rows.append(cur)
            elif cur is not None:
                cur[m.group(1).split(" ")[0]] = m.group(2)
        if not rows:
            continue
        print("## " + os.path.basename(path))
        for r, name in zip(rows, demangle([r["name"] for r in rows])):
            m = re.match(r"_ZN3ucd12_GLOBAL__N_1\d+([a-z0-9_]+_kernel)I(.*?)EEv", name)     # c++filt does not know DF16_ (_Float16)
            if m:
                args = re.findall(r"L([ib])(\d+)E", m.group(2))
                name = m.group(1) + "<" + ", ".join(("true" if v == "1" else "false") if t == "b" else v for t, v in args) + ">"
            m = re.match(r"_ZN3ucd12_GLOBAL__N_1\d+([a-z0-9_]+_kernel)E", name)
            if m:
                name = m.group(1)
            name = re.sub(r"\(anonymous namespace\)::", "", name)
            name = re.sub(r"\(.*$", "", name).replace("void ", "").replace("ucd::", "")
            print("%-5s %-5s %-5s %-8s %-4s %-8s  %s" % (r.get("VGPRs", "?"), r.get("AGPRs", "?"), r.get("TotalSGPRs", "?"),
                                                      r.get("ScratchSize", "?"), r.get("Occupancy", "?"), r.get("LDS", "?"), name))


if __name__ == "__main__":
    main()
