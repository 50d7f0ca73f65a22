import sys, time
sys.path.insert(0, '.')
import iterative_solvers_amd as isa
for N in (256, 1024):
    s = isa.GridSystem(N, N, 1.0, 2.0, 1.0, 2.0); h = s._handle
    for rule, name in ((1, 'rel2'), (0, 'msg')):
        for use_u in (0, 1):
            for sync in (100, 500):
                p = isa.default_params(rule); p.max_iterations = 2000; p.fixed_iterations = 1; p.use_true_solution = use_u; p.sync_every = sync; p.callback_every = 0
                h.solve(p)
                t0 = time.perf_counter(); r = h.solve(p); dt = time.perf_counter() - t0
                print(N, name, 'use_u', use_u, 'sync', sync, f'{dt/2000*1e6:.1f} us/iter', flush=True)
