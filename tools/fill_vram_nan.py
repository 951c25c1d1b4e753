import torch
bl=[]
try:
    for i in range(1000):
        t=torch.empty(256*1024*1024,dtype=torch.float32,device="cuda:0"); t.fill_(float("nan")); bl.append(t)
except RuntimeError as e:
    pass
torch.cuda.synchronize(); print("filled", len(bl), "GiB")
