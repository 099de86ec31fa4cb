import sys
sys.path.insert(0, '.')
def maps(tag):
    libs = sorted({l.split()[-1] for l in open('/proc/self/maps') if 'amdhip' in l or 'hsa-runtime' in l})
    print(tag, libs, flush=True)
order = sys.argv[1]
if order == 'ours_first':
    from sift_amd.texture import TextureContext
    ctx = TextureContext(0)
    maps('after ours')
    import torch
    maps('after import torch')
    try:
        print(torch.cuda.is_available(), torch.zeros(3, device='cuda'))
    except Exception as e:
        print('torch failed:', e)
    maps('end')
else:
    import torch
    print(torch.cuda.is_available(), torch.zeros(3, device='cuda'))
    maps('after torch')
    from sift_amd.texture import TextureContext
    ctx = TextureContext(0)
    maps('after ours')
