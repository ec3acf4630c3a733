#!/usr/bin/env python3
"""Print the synchronisation / memory / MFMA outline of one kernel from a hipcc -S listing.
usage: asm_outline.py file.s <substring of mangled kernel name>"""
import re, sys
s = open(sys.argv[1]).read()
name = sys.argv[2]
m = re.search(r"^(\S*" + re.escape(name) + r"\S*):[^\n]*\n(.*?)s_endpgm", s, re.S | re.M)
if not m:
    sys.exit("kernel not found")
body = m.group(2)
print(m.group(1), "| global_store:", body.count("global_store"), "| glds:", body.count("global_load_lds"),
      "| ds_read:", len(re.findall(r"\bds_read", body)), "| mfma:", body.count("v_mfma"))
KEYS = ["s_barrier", "s_waitcnt vmcnt", "s_waitcnt lgkmcnt", "global_load_lds", "global_store", "global_load_dword", "s_cbranch", "v_mfma", "ds_read", "ds_write", "buffer_"]
out, run, runk = [], 0, None
def flush():
    global run, runk
    if run:
        out.append(f"      {runk} x{run}")
    run, runk = 0, None
for i, l in enumerate(body.split("\n")):
    t = l.strip()
    key = next((k for k in KEYS if t.startswith(k)), None)
    if key is None or key == "s_waitcnt lgkmcnt":
        continue
    if key in ("v_mfma", "ds_read", "global_load_lds", "global_store", "global_load_dword", "ds_write"):
        if runk == key:
            run += 1
        else:
            flush(); runk, run = key, 1
        continue
    flush()
    out.append(f"{i}: {t.split(';')[0][:70]}")
flush()
print("\n".join(out))
