import json, sys
d = json.load(open(sys.argv[1]))
c = d["config"]
print(d["n_gpus"], round(d["ms_per_step"], 2), d["selfcheck"][:100])
print(c["path"]); print(c["library_path"]); print(c.get("relay_trial")); print(c["halo_routing"], c["shard_mode"])
