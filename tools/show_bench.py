import json, sys
for f in sys.argv[1:]:
    try:
        r = json.load(open(f))
    except Exception as e:
        print(f, "unreadable:", e); continue
    print(f, "ms/step %.3f value %.4g" % (r["ms_per_step"], r["value"]))
    if "kernels" in r:
        print("  ", {k: (round(v["us"], 1), round(v["GBps"]), round(v.get("TFLOPs", 0.0), 2)) for k, v in r["kernels"].items()})
        ko = r.get("kernel_only")
        if ko:
            print("   kernel-only us %.1f seq-steps/s %.4g valu_frac %.4f" % (ko["us_per_step"], ko["sequence_steps_per_s"], ko["valu_frac"]))
        print("   roofline", r["roofline"]["kernel"], "frac %.4f" % r["roofline"]["frac"])
    if "cpu_baseline" in r:
        print("   cpu", r["cpu_baseline"]["value"], r["cpu_baseline"]["cores"])
