bash tools/profile_round3.sh r03b > gpurun_out/profile_r03b.log 2>&1
tail -12 gpurun_out/profile_r03b.log
bash tools/instr_budget.sh run > gpurun_out/instr_budget_r03b.txt 2>&1
cat gpurun_out/instr_budget_r03b.txt
