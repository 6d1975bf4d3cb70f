"""Static look at k_cbf_rollout<float, 0, false, 8> in the current sources: registers, spills, instruction mix per stage.
python3 profiles/tools/isa_roll.py   (compiles multidronesim_amd/csrc/mds_api.hip to build/isa/mds_api.s: ~2 minutes)"""
import collections, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
out = os.path.join(ROOT, "build", "isa")
os.makedirs(out, exist_ok=True)
asm = os.path.join(out, "mds_api.s")
if "--no-build" not in sys.argv:
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-Wno-pass-failed", "-w", "-DMDS_PART=2", "-S", "--cuda-device-only",
                           "-o", asm, os.path.join(ROOT, "multidronesim_amd", "csrc", "mds_api.hip")])
s = open(asm).read()
for m in re.finditer(r"\.name:\s+(\S*k_cbf_rollout\S*)", s):
    seg = s[m.start() - 1500:m.start() + 1500]
    g = lambda k: (re.search(r"\." + k + r":\s+(\d+)", seg) or [None, "?"])[1]
    print(m.group(1)[:48], "vgpr", g("vgpr_count"), "spill", g("vgpr_spill_count"), "sgpr", g("sgpr_count"), "lds", g("group_segment_fixed_size"))
lines = s.splitlines()
name = "_ZN3mds13k_cbf_rolloutIfLi0ELb0ELi8ELb0EEE"
st = [k for k, l in enumerate(lines) if l.startswith(name) and "@function" not in l and ": " in l][0]
en = next(k for k in range(st, len(lines)) if "s_endpgm" in lines[k])
body = lines[st + 1:en]
open(os.path.join(out, "roll_f0.s"), "w").write("\n".join(body))
marks = [k for k, l in enumerate(body) if "s_barrier" in l or "ds_add_rtn" in l or "s_setprio" in l]
print(len(body), "lines;", [(k, body[k].strip()) for k in marks])
print("scratch ops at", [k for k, l in enumerate(body) if "scratch_" in l])
def mix(a, b, tag):
    c = collections.Counter(l.strip().split()[0] for l in body[a:b] if l.strip() and not l.strip().startswith((".", ";")) and not l.strip().endswith(":"))
    print(tag, a, b, "valu", sum(n for k, n in c.items() if k.startswith("v_")), "salu", sum(n for k, n in c.items() if k.startswith("s_")),
          "readlane", c["v_readlane_b32"], "writelane", c["v_writelane_b32"], "mov", c["v_mov_b32_e32"], "lshl_add_u64", c["v_lshl_add_u64"], "nop", c["s_nop"])
bars = [k for k in marks if "s_barrier" in body[k]][-2:]
mix(bars[0], bars[1], "B (all variants, solver included)")
mix(bars[1], len(body), "C + A")
