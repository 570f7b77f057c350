set -e
bash tools/ab_check.sh
timeout -k 10 200 python -u tools/fuzz_parity.py --seconds 100 --seed 23 > gpurun_out/fuzz23.txt 2>&1
timeout -k 10 400 python tools/config5_stress.py --spp 64 > gpurun_out/config5.json 2> gpurun_out/config5.err
