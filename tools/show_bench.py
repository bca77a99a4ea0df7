import json
import sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["ms_per_step"], d.get("per_instance_steps"))
