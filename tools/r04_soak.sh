# GPU soak of round 4 (profiles/soak_r04.txt): random legal configurations x all psy models x mixed signals against the oracle, on the
# round's kernels (psy 2/4 as self-seeding runs of frames, mono streams in pairs, the rebuilt threshold walk).  usage: bash tools/r04_soak.sh
mkdir -p gpurun_out/r04soak
: > gpurun_out/r04soak/soak.txt
for seed in 611 612 613 614 615 616 617 618; do timeout 900 python3 tools/soak_gpu.py 8192 12 $seed >> gpurun_out/r04soak/soak.txt 2>&1; done
for seed in 621 622 623 624 625 626; do TL_SOAK_MODELS=2,4 timeout 900 python3 tools/soak_gpu.py 8192 24 $seed >> gpurun_out/r04soak/soak.txt 2>&1; done
for seed in 631 632 633 634; do TL_SOAK_MODELS=1,3 timeout 900 python3 tools/soak_gpu.py 16384 9 $seed >> gpurun_out/r04soak/soak.txt 2>&1; done
for seed in 641 642 643 644; do TL_SOAK_EDGE=1 TL_SOAK_MODELS=2,4,1 timeout 900 python3 tools/soak_gpu.py 8192 12 $seed >> gpurun_out/r04soak/soak.txt 2>&1; done
grep -c "0 mismatching" gpurun_out/r04soak/soak.txt; grep -v "0 mismatching" gpurun_out/r04soak/soak.txt | head
