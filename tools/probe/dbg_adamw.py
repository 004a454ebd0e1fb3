import sys; sys.path.insert(0,'/root/repo')
import torch
from mobilenet_yolo_pytorch_amd.optim import AdamW
def params(seed):
    g = torch.Generator().manual_seed(seed)
    shapes = [(32, 3, 3, 3), (96,), (75, 512, 1, 1), (1,), (7,), (300, 1000), (1280, 320, 1, 1)]
    return [torch.nn.Parameter(torch.randn(*s, generator=g).cuda()) for s in shapes]
pa, pb = params(2), params(2)
oa, ob = torch.optim.AdamW(pa, lr=1e-3), AdamW(pb, lr=1e-3)
g = torch.Generator().manual_seed(3)
for it in range(3):
    for a, b in zip(pa, pb):
        gr = torch.randn(*a.shape, generator=g).cuda()
        a.grad, b.grad = gr.clone(), gr.clone()
    oa.step(); ob.step()
    torch.cuda.synchronize()
    print('step', it, max(float((a-b).abs().max()) for a,b in zip(pa,pb)))
    if it == 0:
        sd = oa.state_dict()
        print({k: (type(v), getattr(v,'device',None)) for k,v in sd['state'][0].items()}, sd['param_groups'][0].keys())
        ob.load_state_dict(sd)
        st = ob.state[pb[0]]
        print('after load', {k: (v.dtype, v.device, float(v.flatten()[0])) for k,v in st.items()}, ob.param_groups[0]['lr'], ob.param_groups[0]['betas'], ob.param_groups[0]['weight_decay'])
        print('moment diff', float((st['exp_avg'] - oa.state[pa[0]]['exp_avg']).abs().max()))
