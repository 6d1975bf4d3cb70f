"""Register / LDS / scratch use of the kernels in multidronesim_amd/libmds.so (reads the gfx950 code object out of the fat binary).
python3 profiles/tools/kstats.py [substring of the mangled kernel name ...]"""
import os, re, struct, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
so = os.path.join(ROOT, "multidronesim_amd", "libmds.so")
with tempfile.TemporaryDirectory() as td:
    fat = os.path.join(td, "fat.bin")
    subprocess.check_call(["/opt/rocm/lib/llvm/bin/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", so])
    data = open(fat, "rb").read()
    n = struct.unpack_from("<Q", data, 24)[0]
    off = 32
    co = None
    for _ in range(n):
        o, sz, tl = struct.unpack_from("<QQQ", data, off)
        off += 24
        tr = data[off:off + tl].decode()
        off += tl
        if "gfx950" in tr:
            co = os.path.join(td, "dev.co")
            open(co, "wb").write(data[o:o + sz])
    notes = subprocess.check_output(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", co], text=True)
    filt = subprocess.Popen(["c++filt"], stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True)
cur = {}
rows = []
for ln in notes.splitlines():
    m = re.match(r"\s+-?\s*\.(\w+):\s+(.*)", ln)
    if not m:
        continue
    k, v = m.group(1), m.group(2).strip()
    if k == "name" and v.startswith("_Z"):
        cur["name"] = v
    if k in ("vgpr_count", "sgpr_count", "private_segment_fixed_size", "group_segment_fixed_size", "vgpr_spill_count", "agpr_count"):
        cur[k] = v
    if k == "wavefront_size":
        if "name" in cur:
            rows.append(cur)
        cur = {}
names, _ = filt.communicate("\n".join(r["name"] for r in rows))
for r, dn in zip(rows, names.splitlines()):
    if sys.argv[1:] and not any(a in r["name"] or a in dn for a in sys.argv[1:]):
        continue
    print(f"vgpr {r.get('vgpr_count'):>4} sgpr {r.get('sgpr_count'):>4} lds {r.get('group_segment_fixed_size'):>7} scratch {r.get('private_segment_fixed_size'):>5} spill {r.get('vgpr_spill_count'):>3}  {dn[:150]}")
