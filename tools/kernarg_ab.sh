# A/B of the HIP runtime's kernarg placement (HIP_FORCE_DEV_KERNARG) and graph packet capture on the bench workload
run() { python bench.py --steps 40 --warmup 5 --no-cpu-baseline --profile-steps 0 --small-batch-steps 20 --fp32-steps 0 --f16x3-steps 0 2>&1 | grep -o '"value": [0-9.]*\|"eager_steps_per_s": [0-9.]*\|"graph_steps_per_s": [0-9.]*' | tr '\n' ' '; echo; }
for rep in 1 2; do
  echo -n "unset: "; run
  echo -n "HIP_FORCE_DEV_KERNARG=1: "; HIP_FORCE_DEV_KERNARG=1 run
  echo -n "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0: "; DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 run
  echo -n "DEBUG_CLR_GRAPH_PACKET_CAPTURE=1: "; DEBUG_CLR_GRAPH_PACKET_CAPTURE=1 run
done
