import sys; sys.path.insert(0, "/root/repo")
import torch
from detectinblur_amd import _lib
n = 200000
x = (torch.rand(n, device="cuda") * 2 + 0.2)
one = torch.ones(n, device="cuda"); zero = torch.zeros(n, device="cuda")
w = torch.ones(n, 1, device="cuda")
wf = torch.empty_like(w); sc = torch.empty(n, device="cuda"); sh = torch.empty(n, device="cuda")
P = _lib.ptr_array
_lib.check(_lib.lib().dib_fold_bn_multi(P([w.data_ptr()]), P([one.data_ptr()]), P([zero.data_ptr()]), P([zero.data_ptr()]), P([x.data_ptr()]),
                                        _lib.int_array([n]), _lib.int_array([1]), 1, 0.0, P([wf.data_ptr()]), P([sc.data_ptr()]), P([sh.data_ptr()]), 0))
torch.cuda.synchronize()
a = torch.rsqrt(x); b = 1 / torch.sqrt(x); c = torch.sqrt(1 / x)
print("kernel == torch.rsqrt:", int((sc != a).sum()), " kernel == 1/sqrt:", int((sc != b).sum()), " rsqrt == 1/sqrt:", int((a != b).sum()))
d = (x.double().rsqrt()).float()
print("torch.rsqrt == correctly rounded:", int((a != d).sum()), " kernel == correctly rounded:", int((sc != d).sum()), " 1/sqrt == cr:", int((b != d).sum()))
