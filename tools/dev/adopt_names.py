import sys; sys.path.insert(0,'.')
import torch, pepflowww_amd
from pepflowww_amd import synth, backward as BW, train_step as TS
from pepflowww_amd.train_forward import default_train_noise
dev=torch.device('cuda:0'); B,L=16,128
model=pepflowww_amd.FlowModel(pepflowww_amd.default_config()); model.load_state_dict(synth.seeded_state_dict()); model=model.to(dev).train()
batch={k:v.to(dev) for k,v in synth.make_pocket_batch(B,L,16,seed=114514).items()}
gen=torch.Generator().manual_seed(1234)
names, sd = TS._state_dict_f32(model)
nz={k:v.to(dev,torch.float32).contiguous() for k,v in default_train_noise(B,L,gen).items()}
wdev=torch.tensor([0.5,0.5,0.25,1,1,0.5],dtype=torch.float32,device=dev)
seed_dev=torch.zeros(1,dtype=torch.int64,device=dev)
orig=BW.GradArena.adopt
def adopt(self, grads):
    for n,g in grads.items():
        if g is not None and not self.owns(g): print('ADOPT', n, tuple(g.shape))
    return orig(self, grads)
BW.GradArena.adopt=adopt
oz=BW._zeros
import traceback
def zz(*shape, device, dtype=torch.float32):
    f=[x for x in traceback.extract_stack() if 'pepflowww_amd' in x.filename][-1]
    print('ZEROS', shape, f.filename.split('/')[-1], f.lineno)
    return oz(*shape, device=device, dtype=dtype)
BW._zeros=zz
losses,state=TS._step_forward(model,sd,batch,nz,0,0,seed_dev=seed_dev)
grads,arena=TS._step_backward(state,wdev,return_arena=True)
torch.cuda.synchronize()
