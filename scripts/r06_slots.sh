#!/bin/bash
# round 6: slots per packed launch after the tap changes -- 40 / 48 / 56 / 64 on the 200- and 400-utterance egs/mol jobs
for J in 200 400; do for G in 40 48 56 64; do
  echo -n "egs/mol job $J, $G slots: "; python bench.py --job $J --packed --job-group $G --cpu-steps 0 --no-extras 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.readlines()[-1]); j=d["job"]; print(d["value"], round(j["padding_loss"],4), j["rank0_launches_B_x_T"])'
done; done
for G in 24 32 40; do
  echo -n "cfg4 job 128, $G slots: "; python bench.py --workload cfg4_mol_multispeaker --job 128 --packed --job-group $G --cpu-steps 0 --no-extras 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.readlines()[-1]); j=d["job"]; print(d["value"], round(j["padding_loss"],4), j["rank0_launches_B_x_T"])'
done
