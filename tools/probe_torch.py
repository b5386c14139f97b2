import sys, time
sys.path.insert(0, 'image-processing_amd'); sys.path.insert(0, '.')
t=time.time()
import torch
print('torch', torch.__version__, 'import %.1fs' % (time.time()-t), torch.cuda.is_available())
import numpy as np
import lanecv
from lanecv import synth, _abi
from lanecv.pipeline import LanePipeline, Recipe
w,h=960,540
rec=Recipe.north_star(w,h)
pipe=LanePipeline(w,h,rec,max_batch=4,nslots=1)
fr=np.stack([synth.synth_road(w,h,1,t) for t in range(4)])
d=torch.from_numpy(fr).cuda()
torch.cuda.synchronize()
pipe.submit_device(0, d.data_ptr(), 4)
a=pipe.collect(0)
pipe.submit(0, fr)
b=pipe.collect(0)
print('same', all((x==y).all() for x,y in zip(a,b)), [len(x) for x in a])
import subprocess
print(subprocess.run("grep -E 'libamdhip64|liblanecv' /proc/%d/maps | awk '{print $6}' | sort -u" % __import__('os').getpid(), shell=True, capture_output=True, text=True).stdout)
