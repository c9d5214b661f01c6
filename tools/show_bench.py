"""Development tool: the numbers of a bench.py JSON line that matter while tuning."""
import json, sys
for f in sys.argv[1:]:
    d = json.load(open(f))
    st = d.get("roofline", {}).get("stage_ms", {})
    k = d["steps"]
    print(f, f'{d["value"]:.0f} Mrays/s {d["samples_per_s"] / 1e6:.1f} Msamples/s | per step ms: ' + " ".join(f"{a} {b / k:.1f}" for a, b in st.items()),
          "| parity", (d.get("parity") or {}).get("pass"), "| f64", (d.get("f64") or {}).get("samples_per_s"))
