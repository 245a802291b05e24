"""Dev tool: the process-level end-to-end figures of a bench line on stdin.  usage: python bench.py --no-ingest --no-cpu-baseline | python profiles/tools/e2e_show.py"""
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=d["e2e"]
print(e["process_wall_s_all_runs"], e["process_wall_s_back_to_back"])
e2=d["e2e_config2"]; print(e2["process_wall_s_all_runs"], e2.get("host_over_device"), e2.get("device_s"))
for sg in e2.get("segments_s_all_runs", [])[:2]: print({k:round(v,3) for k,v in sg.items() if v>0.003})
