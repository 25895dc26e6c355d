set -e
mkdir -p gpurun_out/sweep
timeout -k 10 300 python -u -m pytest tests/test_gpu_lstm.py -x -q --timeout 120 --timeout-method thread -k "layer_step" > gpurun_out/sweep/lstm_step_test.log 2>&1
for T in 8 20 36; do for L in 16 4; do
KVAE_TIME_T=$T KVAE_SEQ_LANES=$L timeout -k 10 120 python -u tools/time_loss_kernels.py 2048 4096 5120 6144 8192 12288 | grep value_and_grad >> gpurun_out/sweep/row_vs_quad.log
done; done
