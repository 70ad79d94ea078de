import json, sys
d = json.load(open(sys.argv[1]))
r = d["roofline"]
print("value %.1f %s | %.3f ms/step | kernel %s %.3f ms | layout %.3f ms | frac %.4f (step %.4f) | err %s" % (
    d["value"], d["unit"], d["ms_per_step"], r["kernel"], r["kernel_ms"], r["layout_pass_ms"], r["frac"], r["step_frac"], d.get("max_abs_vs_ref")))
