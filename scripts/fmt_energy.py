import sys, json
for l in sys.stdin:
    if l.startswith("{"):
        d = json.loads(l)
        print(d["bp"], d["dtype"], *["%s: %.2f ms" % (k, d[k]["ms_per_call"]) for k in ("energy", "energy+forces", "energy+forces+dU/dtheta")])
