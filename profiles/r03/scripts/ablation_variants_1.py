import os, shutil, subprocess, sys
src = "/root/repo/fictitious_domain_al_preconditioners_amd/csrc"
base = open(f"{src}/kernels_vs.hpp").read()
def variant(n, edits):
    d = f"/tmp/abl/v{n}"
    shutil.rmtree(d, ignore_errors=True)
    shutil.copytree(src, d)
    s = base
    for a, b in edits:
        assert a in s, (n, a)
        s = s.replace(a, b)
    open(f"{d}/kernels_vs.hpp", "w").write(s)
    out = f"/root/repo/scratch/abl/libalfd_v{n}.so"
    cmd = f"/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -w -I/root/repo/include -o {out} {d}/alfd.hip -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib"
    return subprocess.Popen(cmd, shell=True)
X8 = "for (int i = 0; i < R; ++i) xv[i] = vs_lds_f64(VsFmt<WD>::kWinOff + (uint32_t)((int32_t)lc + sh[i]));"
ps = []
# 1: no tree (sum of the lane's accumulators)
ps.append(variant(1, [("const double s8 = reduce_rows8(acc[0], acc[1], acc[2], acc[3], acc[4], acc[5], acc[6], acc[7], lane);",
                       "const double s8 = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));")]))
# 2: no x gathers in shared batches (x taken from a register)
ps.append(variant(2, [(X8, "for (int i = 0; i < R; ++i) xv[i] = __hiloint2double((int)lc + sh[i], (int)lc);")]))
# 3: conflict-free x gathers (lane-private bank), adds kept
ps.append(variant(3, [(X8, "for (int i = 0; i < R; ++i) xv[i] = vs_lds_f64(VsFmt<WD>::kWinOff + (uint32_t)((((int32_t)lc + sh[i]) & 0x1e00) | (lane << 3)));")]))
# 4: no window staging
ps.append(variant(4, [("    for (int32_t s = s0 + wave * U; s < s1; s += NW * U) {", "    for (int32_t s = s0 + wave * U; s < s1 && nd < 0; s += NW * U) {")]))
# 5: no dictionary gather (value from register)
ps.append(variant(5, [("    double v = vs_lds_f64(kVsDictOff + ((w[j] >> VsFmt<WD>::kDictShift) & VsFmt<WD>::kDictMask));\n    double xv[R];",
                       "    double v = __hiloint2double((int)(w[j] >> 3), (int)w[j]);\n    double xv[R];")]))
for p in ps:
    p.wait()
    print("rc", p.returncode)
