"""Prepend the provenance header to the round-2 profile summaries copied from gpurun_out/r02 (scripts/gpu_profiles_r02.sh)."""
import os
P = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'profiles')
HDR = {
    'r02_train_step_kernels.txt': '# rocprofv3 --kernel-trace --stats -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary   (MI355X, round 2; 3 train steps +\n# the 4 decoder forwards of the step-roofline leg in the trace).  Per (kernel, workgroups, HSA queue) table by scripts/trace_summary.py:\n# queue 1 = caller stream (attention chain), 2 = side stream (generator chain, per-chunk input-gradient GEMMs), 3 = weight-gradient stream.\n# lstm_gates_kernel<0,4> (416 wg, q1) = attention-LSTM gate GEMM, (256 wg, q2) = generator-LSTM; lstm_cell_q_kernel = cell (+ query partials);\n# gemm_pipe_kernel<TA,TB,CONV> = software-pipelined split-bf16 GEMM core (CONV 1/2/3 = convolution forward / input gradient / weight gradient).\n# rocprofv3\'s own per-kernel statistics of the same run: r02_train_step_kernel_stats.csv\n',
    'r02_train_step_phases.txt': '# phases of the last traced train step of the same run (scripts/phase_summary.py); the profiler stretches the step from ~86 to ~100 ms\n',
    'r02_fwd_decoder_trace.txt': '# rocprofv3 --kernel-trace -- python bench.py --traffic-probe --preset shared_training --batch 64: the 240-frame teacher-forced decoder forward\n# between two mtts_marker_kernel launches (scripts/trace_summary.py --region 2).  Last line: how little the two chains overlap.\n',
    'r02_pmc_hbm_traffic.txt': '# rocprofv3 --kernel-trace --pmc FETCH_SIZE (pass 1) / --pmc WRITE_SIZE (pass 2) -- python bench.py --traffic-probe ...: per-launch averages by\n# kernel / grid / HSA queue over the decoder FORWARD (scripts/pmc_summary.py).  gfx950: FETCH_SIZE counts 64 B per 128-B request of wide\n# coalesced reads -> x2.  bench.py measures the per-step total live (roofline.traffic) from the same two passes.\n',
    'r02_pmc_train_step_traffic.txt': '# the same two PMC passes over python bench.py --steps 1 --warmup 1 --frames 96 (forward AND backward kernels, 2 train steps + step-roofline leg).\n# attn_bwd_plus_skinny_kernel: 58.8 MB fetched + 9.4 MB written per launch; backward skinny products (queue 1: 121 856 threads = ctx-columns,\n# 32 768 threads = cell backward; queue 2: generator chain).\n',
    'r02_generated_training_kernels.txt': '# rocprofv3 --kernel-trace -- python bench.py --steps 1 --warmup 1 --preset generated_training --batch 60 (2 train steps + 4 decoder forwards):\n# gen_params_fwd/bwd_kernel = K2, the generator writing the conv layout directly.  Bench line: r02_generated_training_line.json\n',
    'r02_inference_kernels.txt': '# rocprofv3 --kernel-trace -- python scripts/prof_inference.py --frames 240: batched synthesis, 128 utterances x 201 tokens, region = one inference_batch call\n',
}
for f, h in HDR.items():
    path = os.path.join(P, f)
    if os.path.exists(path):
        s = open(path).read()
        if not s.startswith('#'):
            open(path, 'w').write(h + s)
