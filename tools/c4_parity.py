import sys, os, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import svi_mapper_amd as svi
from svi_mapper_amd import synth
from oracle import oracle
import importlib.util
spec=importlib.util.spec_from_file_location('t', 'tests/test_ba_gpu.py'); t=importlib.util.module_from_spec(spec); spec.loader.exec_module(t)
prob = synth.make_c4()
g,_ = t._make(svi.BundleAdjuster, prob); o,_ = t._make(oracle.OracleBA, prob)
g.initialize(); o.initialize()
for n in (1,2):
    g.optimize(n); o.optimize(n)
Tg=g.get_poses()[1]; To=o.get_poses()[1]; pg=g.get_landmarks()[1]; po=o.get_landmarks()[1]
print("translation rel %.3e rotation abs %.3e landmarks rel %.3e chi2 rel %.3e" % (t._rel(Tg[:,9:],To[:,9:]), np.abs(Tg[:,:9]-To[:,:9]).max(), t._rel(pg,po), abs(g.last_plain_chi2-o.last_plain_chi2)/o.last_plain_chi2))
