"""Static VALU instruction mix of every kernel in ssvio_amd/libssvio_gpu.so, by the two issue classes measured on this chip
(profiles/r2/ubench_valu_table.txt: ~2.4 cycles per wave-instruction per SIMD for and / or / xor / add / sub / right shifts /
mov / f32 add-sub-mul and the 16-bit VOP2 forms, ~4.2 for everything else, ~5 for v_cmp), so that a kernel's SQ_INSTS_VALU
can be priced in issue cycles instead of a flat 4 per instruction.  Static counts over the kernel's code, not a dynamic
profile: loops weigh as much as prologues.  Writes profiles/r3/valu_mix.json.   python tools/valu_mix.py"""
import json, os, re, subprocess, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "ssvio_amd", "libssvio_gpu.so")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
FAST = re.compile(r"^v_(and|or|xor|not)_b32|^v_(add|sub|subrev)(_co)?_u32|^v_(add|sub|subrev)_(f32|i32)|^v_mul_f32|^v_(lshrrev|ashrrev)_(b32|i32)|^v_mov_b32"
                  r"|^v_(min|max|add|sub)_[iu]16|^v_mul_lo_u16|^v_accvgpr")
def cycles(m):
    if m.startswith("v_cmp") or m.startswith("v_cmpx"): return 5.0
    if m.startswith("v_fma_f32") or m.startswith("v_fmac_f32"): return 2.8
    return 2.4 if FAST.match(m) else 4.2
def main():
    tmp = "/tmp/ssvio_offload"
    os.makedirs(tmp, exist_ok=True)
    # the fat binary sits in the .hip_fatbin section of the shared object; llvm-objdump --offloading writes the code objects
    # beside its input, so it works on a copy
    import shutil
    shutil.copy(SO, os.path.join(tmp, "lib.so"))
    subprocess.check_call([OBJDUMP, "--offloading", "lib.so"], cwd=tmp, stdout=subprocess.DEVNULL)
    out = {}
    for f in sorted(os.listdir(tmp)):
        if "gfx950" not in f: continue
        dis = subprocess.run([OBJDUMP, "-d", os.path.join(tmp, f)], capture_output=True, text=True).stdout
        cur = None
        for line in dis.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
            if m:
                cur = m.group(1); continue
            t = line.split()
            if cur is None or not t or not t[0].startswith(("v_", "s_", "ds_", "global_", "buffer_", "flat_")): continue
            if t[0].startswith("v_") and not t[0].startswith(("v_mfma", "v_smfmac")):
                d = out.setdefault(cur, collections.Counter())
                d["n"] += 1; d["cyc"] += cycles(t[0]); d["fast" if cycles(t[0]) == 2.4 else "slow"] += 1
            elif t[0].startswith("v_mfma"):
                out.setdefault(cur, collections.Counter())["mfma"] += 1
    res = {}
    for k, d in out.items():
        name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"^void ", "", name).replace("ssvio::", "").split("(")[0]
        if d["n"]:
            res[name] = {"valu_static": d["n"], "fast_class": d["fast"], "avg_issue_cycles": round(d["cyc"] / d["n"], 3), "mfma_static": d.get("mfma", 0)}
    os.makedirs(os.path.join(ROOT, "profiles", "r3"), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, "profiles", "r3", "valu_mix.json"), "w"), indent=1, sort_keys=True)
    for k in sorted(res): print("%-48s %6d valu  %.2f cyc/instr  %d mfma" % (k[:48], res[k]["valu_static"], res[k]["avg_issue_cycles"], res[k]["mfma_static"]))
main()
