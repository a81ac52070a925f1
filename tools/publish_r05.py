#!/usr/bin/env python3
"""gpurun_out/rend/ (tools/gpu_round_end.sh a + b on the GPU box) -> profiles/r05_* (tracked): the bench record, the timed-only kernel
summary, the PMC passes, the small configurations, the cooperative step's sweep and stamps, the one-launch Cholesky's sections."""
import shutil

R, P = 'gpurun_out/rend/', 'profiles/r05_'
clean = lambda path: "".join(l for l in open(path) if "amdgpu.ids" not in l and "run_backward" not in l and "UserWarning" not in l)
for a, b in (('bench_C3.json', 'bench_C3.json'), ('bench_C3_timed_only_kernel_summary.md', 'bench_C3_timed_only_kernel_summary.md'),
             ('bench_C3_timed_only_kernel_stats.csv', 'bench_C3_timed_only_kernel_stats.csv'),
             ('bench_C3_profiled_timed_only.json', 'bench_C3_timed_only_profiled_run.json'), ('pmc_gemm.json', 'pmc_gemm.json')):
    shutil.copy(R + a, P + b)
old = open(P + 'small_configs.txt').read()
i = old.index('\n# tools/bo_iteration_mid.py')
open(P + 'small_configs.txt', 'w').write("# the small configurations through the one-launch steps and through the layer path (bench.py, 300 steps x 5 repeats, MI355X)\n"
                                         + clean(R + 'bench_small_configs.txt') + old[i:])
old = open(P + 'coop_step.txt').read()
i_cond, i_st, i_ab = old.index('# tools/cond_bench_mid.py'), old.index('# tools/coop_stamps.py'), old.index('# A/B, same box: the register Cholesky')
sweep = "# tools/coop_sweep.py: the cooperative one-launch step (mobocmf_coop_elbo_step) against the layer path, us per launch of the whole group\n" + clean(R + 'coop_sweep.txt')
st_head = old[i_st:old.index('\n', i_st) + 1]
open(P + 'coop_step.txt', 'w').write(sweep + "\n" + old[i_cond:i_st] + st_head + clean(R + 'coop_stamps.txt') + "\n" + old[i_ab:])
s = open(P + 'chol_one_launch.txt').read()


def between(a, b):
    i = s.index(a)
    return i, s.index(b, i)


h1 = "substitution: 1-3 ulp more backward error at cond ~ 5e8, still at rocSOLVER's level)\n"
i, j = between(h1, '#\n# == 2.')
s = s[:i] + h1 + clean(R + 'chol_accuracy.txt') + s[j:]
h2 = 'matrix-vector product, the likelihood; eager launches, 50 calls)\n'
i, j = between(h2, '#\n# == 3.')
s = s[:i] + h2 + clean(R + 'chol_bench.txt') + s[j:]
i = s.index('potrf_cols 0  --config C3 ')
j = s.index('#       (', i)      # the comment block behind the six lines
s = s[:i] + open(R + 'chol_ab.txt').read() + s[j:]
open(P + 'chol_one_launch.txt', 'w').write(s)
print("published")
