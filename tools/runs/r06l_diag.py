import sys, torch
sys.path.insert(0, "/root/repo")
import tests.test_gpu_bn_block as t
for dtype in (torch.bfloat16, torch.float16, torch.float32):
    for shape in ((24, 16, 14, 14), (64, 16, 14, 14)):
        tr, fu = t.run(shape, dtype, True, 7 + shape[1], fused=True)
        _, un = t.run(shape, dtype, True, 7 + shape[1], fused=False)
        for k in (0, 1):
            a, b, c = tr["bn"][k], fu["bn"][k], un["bn"][k]
            print(dtype, shape, "bn grad", k, "scale %.2f" % float(a.abs().max()), "fused-truth %.3f" % float((a - b).abs().max()),
                  "unfused-truth %.3f" % float((a - c).abs().max()), "fused-unfused %.3f" % float((b - c).abs().max()))
        print("   sn grads fused-truth", [round(float((a - b).abs().max()), 4) for a, b in zip(tr["sn"], fu["sn"])],
              "unfused-truth", [round(float((a - b).abs().max()), 4) for a, b in zip(tr["sn"], un["sn"])])
        print("   dc max err fused %.4f unfused %.4f  (scale %.2f)" % (float((tr["dc"] - fu["dc"]).abs().max()), float((tr["dc"] - un["dc"]).abs().max()), float(tr["dc"].abs().max())))
