#!/bin/bash
# Does the host cost of an eager step depend on which NUMA node the process runs on?
python3 - <<'PY'
import torch, glob, os
p = torch.cuda.get_device_properties(0)
bdf = "%04x:%02x:%02x.0" % (getattr(p, "pci_domain_id", 0), p.pci_bus_id, p.pci_device_id)
print("gpu", p.name, bdf)
d = "/sys/bus/pci/devices/" + bdf
for f in ("numa_node", "local_cpulist"):
    try: print(f, open(os.path.join(d, f)).read().strip())
    except Exception as e: print(f, "ERR", e)
for n in sorted(glob.glob("/sys/devices/system/node/node*/cpulist")):
    print(n, open(n).read().strip())
print("affinity now:", len(os.sched_getaffinity(0)), "cpus")
PY
for n in /sys/devices/system/node/node*/cpulist; do
  cpus=$(cat $n)
  echo "== taskset -c $cpus"
  taskset -c $cpus python3 scratch/t_hostcost.py 2>/dev/null | grep -E "^(800|70)"
done
echo "== no pinning"; python3 scratch/t_hostcost.py 2>/dev/null | grep -E "^(800|70)"
