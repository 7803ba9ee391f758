# dev: r04_state_reroll.py in P processes on one box (results -> gpurun_out/state_reroll.txt)
P=${1:-4}; N=${2:-6}
mkdir -p gpurun_out; : > gpurun_out/state_reroll.txt
for i in $(seq $P); do echo "== process $i" >> gpurun_out/state_reroll.txt; timeout 300 python tools/dev/r04_state_reroll.py $N $i >> gpurun_out/state_reroll.txt 2>&1; done
grep -v amdgpu.ids gpurun_out/state_reroll.txt | grep "==\|sequential engines\|four engines"
