"""Mean / max socket power and mean shader clock per leg of tools/profile_round2.sh (rocm-smi JSON lines, 5 Hz)."""
import json
import os
import re
import sys

out = sys.argv[1]
for leg in ("f16x3", "f32", "gemm"):
    p = os.path.join(out, f"smi_{leg}.jsonl")
    if not os.path.exists(p):
        continue
    pw, ck = [], []
    for line in open(p):
        try:
            d = json.loads(line)
        except Exception:
            continue
        card = d.get("card0", {})
        for k, v in card.items():
            if "ower" in k and "W" in k:
                try:
                    pw.append(float(re.sub(r"[^0-9.]", "", str(v))))
                except ValueError:
                    pass
            if k.startswith("sclk"):
                m = re.search(r"(\d+)\s*Mhz", str(v), re.I)
                if m:
                    ck.append(float(m.group(1)))
    busy = [x for x in pw if x > 0.6 * max(pw)] if pw else []
    bi = [i for i, x in enumerate(pw) if busy and x > 0.6 * max(pw)]
    cb = [ck[i] for i in bi if i < len(ck)]
    extra = ""
    for f in (f"bench_{leg}.json", "gemm.log"):
        fp = os.path.join(out, f)
        if os.path.exists(fp) and (leg in f or leg == "gemm" and f == "gemm.log"):
            t = open(fp).read().strip().splitlines()[-1] if open(fp).read().strip() else ""
            if f.endswith(".json") and t.startswith("{"):
                j = json.loads(t)
                extra = f"{j['value']} pairs/s, {j['ms_per_step']} ms/step"
            elif f == "gemm.log":
                extra = t
    if pw:
        print(f"{leg:6s} samples {len(pw):4d}  power while busy: mean {sum(busy) / max(len(busy), 1):7.1f} W  max {max(pw):7.1f} W   "
              f"sclk while busy: mean {sum(cb) / max(len(cb), 1):7.1f} MHz   {extra}")
