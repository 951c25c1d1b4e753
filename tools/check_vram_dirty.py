import torch
t = torch.empty(1024 * 1024 * 1024, dtype=torch.float32, device="cuda:0")      # 4 GiB fresh from the driver
print("fresh 4 GiB: NaN fraction %.4f, non-zero fraction %.4f" % (torch.isnan(t).float().mean().item(), (t != 0).float().mean().item()))
