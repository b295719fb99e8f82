import sys, time; sys.path.insert(0,'/root/repo')
import numpy as np
import prost_amd as prost
from prost_amd import synthetic
import oracle
for prec, dt in (('single', np.float32), ('double', np.float64)):
    prost.set_precision(prec)
    for fused in (True, False):
        for step in ('alg1','alg2','goldstein','boyd'):
            prob,u,q,f = synthetic.rof_problem(40, 64, L=2)
            b = prost.backend.pdhg(stepsize=step, residual_iter=3, alg2_gamma=0.5)
            b[1]['allow_fused'] = fused
            o = prost.options(max_iters=50, num_cback_calls=0, verbose=False)
            s = prost.Solver(prob, b, o); s.iterate(50); st = s.state()
            bo = prost.backend.pdhg(stepsize=step, residual_iter=3, alg2_gamma=0.5)
            so = oracle.Solver(prob.data, prob.nrows, prob.ncols, bo, o, dt); so.initialize(); so.iterate(50); ost = so.state(); osc = so.scalars()
            print(prec, st['path'], step, [float(np.abs(st[k]-ost[k]).max()) for k in 'xyzw'], st['tau']-osc['tau'], st['primal_res']-osc['primal_res'], st['dual_res']-osc['dual_res'])
            s.destroy()
prost.set_precision('single')
prob,u,q,f = synthetic.rof_problem(256,256)
b = prost.backend.pdhg(stepsize='alg2', residual_iter=10, alg2_gamma=0.5)
def cb(it,x,y): print('cb',it, x.sum()); return False
o = prost.options(max_iters=1000, num_cback_calls=4, verbose=True, interm_cb=cb)
t=time.time(); r = prost.solve(prob,b,o); print(r['result'], r['iters'], r['path'], time.time()-t)
ro = oracle.solve(synthetic.rof_problem(256,256)[0], b, prost.options(max_iters=1000,num_cback_calls=0,verbose=False), np.float32)
print(ro['result'], ro['iters'], np.abs(ro['x']-r['x']).max())
