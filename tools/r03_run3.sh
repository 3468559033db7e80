bash tools/profile_round3.sh r03a > gpurun_out/profile_r03a.log 2>&1
tail -30 gpurun_out/profile_r03a.log
bash tools/instr_budget.sh run > gpurun_out/instr_budget_r03a.txt 2>&1
cat gpurun_out/instr_budget_r03a.txt
