import json,sys
b=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0])
p=b["pipeline"]
print("value %.4g frac %.3f" % (b["value"],b["roofline"]["frac"]))
print("pipeline %.3f %.3f" % (p["ms"],p["roofline"]["frac"]), ["%.3f"%x for x in p["ms_calls"]])
print("overlapped %.3f %.3f batched %.3f %.3f" % (p["overlapped"]["ms_per_partition"],p["overlapped"]["roofline"]["frac"],p["batched"]["ms_per_partition"],p["batched"]["roofline"]["frac"]))
print("small %.3f %.3f" % (p["small"]["ms"],p["small"]["roofline"]["frac"]), ["%.3f"%x for x in p["small"]["ms_calls"]])
print("sparse %.3f %.3f batched %.3f" % (p["sparse"]["ms"],p["sparse"]["roofline"]["frac"],p["sparse"]["batched"]["roofline"]["frac"]), ["%.3f"%x for x in p["sparse"]["ms_calls"]])
