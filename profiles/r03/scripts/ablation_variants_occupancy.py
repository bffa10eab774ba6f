import os, shutil, subprocess, sys
src = "/root/repo/fictitious_domain_al_preconditioners_amd/csrc"
def variant(n, file, edits):
    d = f"/tmp/abl/v{n}"
    shutil.rmtree(d, ignore_errors=True)
    shutil.copytree(src, d)
    s = open(f"{d}/{file}").read()
    for a, b in edits:
        assert a in s, (n, a)
        s = s.replace(a, b)
    open(f"{d}/{file}", "w").write(s)
    out = f"/root/repo/scratch/abl/libalfd_v{n}.so"
    cmd = f"/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -w -I/root/repo/include -o {out} {d}/alfd.hip -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib"
    return subprocess.Popen(cmd, shell=True)
L = "  const size_t lds = (size_t)(v.wide ? VsFmt<1>::kWinOff : VsFmt<0>::kWinOff) + (size_t)v.maxW * sizeof(double);"
ps = []
for n, pad in ((11, 8192), (12, 20480), (13, 45056)):
    ps.append(variant(n, "alfd.hip", [(L, L.replace(";", f" + {pad};") + f'\n  static bool once = false; if (!once) {{ once = true; std::fprintf(stderr, "[abl] lds %zu maxW %d\\n", lds, (int)v.maxW); }}')]))
for p in ps:
    p.wait(); print("rc", p.returncode)
