# CPU leg over the bench window on the GPU box's host, in calls of <= 20 min (checkpoint carried in profiles/r02/.ckpt)
mkdir -p gpurun_out/ckpt
LOAD=""; FIRST=0
if [ -f profiles/r02/.ckpt/cpu_window.npz ]; then LOAD="--load profiles/r02/.ckpt/cpu_window.npz"; FIRST=$(python -c "import numpy as np; print(len(np.load('profiles/r02/.ckpt/cpu_window.npz')['its']))"); fi
python tools/cpu_window.py --steps 50 --threads 1 --first $FIRST $LOAD --save gpurun_out/ckpt/cpu_window.npz --time-budget 1050 --out gpurun_out/cpu_window_1thread.json > gpurun_out/cpu_window_b_$FIRST.log 2>&1
tail -2 gpurun_out/cpu_window_b_$FIRST.log
