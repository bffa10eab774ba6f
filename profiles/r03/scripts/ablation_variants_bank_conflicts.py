import os, shutil, subprocess, sys
src = "/root/repo/fictitious_domain_al_preconditioners_amd/csrc"
def variant(n, file, edits):
    d = f"/tmp/abl/v{n}"
    shutil.rmtree(d, ignore_errors=True)
    shutil.copytree(src, d)
    s = open(f"{d}/{file}").read()
    for a, b, c in edits:
        assert s.count(a) >= 1, (n, a)
        s = s.replace(a, b, c)
    open(f"{d}/{file}", "w").write(s)
    out = f"/root/repo/scratch/abl/libalfd_v{n}.so"
    cmd = f"/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -w -I/root/repo/include -o {out} {d}/alfd.hip -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib"
    return subprocess.Popen(cmd, shell=True)
X = "      for (int i = 0; i < G; ++i) xv[i] = vs_lds_f64(VsFmt<WD>::kWinOff + (uint32_t)((int32_t)lc + sh[g + i]));"
ps = []
# x gathers forced onto lane-private banks (16-lane groups hit 16 distinct 8-byte banks)
ps.append(variant(31, "kernels_vs.hpp", [(X, "      for (int i = 0; i < G; ++i) xv[i] = vs_lds_f64(VsFmt<WD>::kWinOff + ((((uint32_t)((int32_t)lc + sh[g + i])) & 0x3f80u) | ((uint32_t)(lane & 15) << 3)));", 1)]))
# the same for the dictionary gather too
ps.append(variant(32, "kernels_vs.hpp", [(X, "      for (int i = 0; i < G; ++i) xv[i] = vs_lds_f64(VsFmt<WD>::kWinOff + ((((uint32_t)((int32_t)lc + sh[g + i])) & 0x3f80u) | ((uint32_t)(lane & 15) << 3)));", 1),
    ("    double v = vs_lds_f64(kVsDictOff + ((wj >> VsFmt<WD>::kDictShift) & VsFmt<WD>::kDictMask));\n    constexpr int G", "    double v = vs_lds_f64(kVsDictOff + ((((wj >> VsFmt<WD>::kDictShift) & VsFmt<WD>::kDictMask) & 0xf80u) | ((uint32_t)(lane & 15) << 3)));\n    constexpr int G", 1)]))
for p in ps:
    p.wait(); print("rc", p.returncode)
