import sys, os
sys.path.insert(0, os.getcwd())
mode = sys.argv[1]
if mode == "lib_first":
    import kmdiff_amd as K
    print("lib device:", K.device_name())
    import torch
    print("torch sees", torch.cuda.device_count()); x = torch.zeros(1, device="cuda"); print("ok", x.device)
else:
    import torch
    x = torch.zeros(1, device="cuda"); print("torch ok")
    import kmdiff_amd as K
    print("lib device:", K.device_name())
