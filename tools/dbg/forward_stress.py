"""Inference-forward stress: shapes that take different kernels (fused GEMM + LayerNorm above 24,576 rows, the unfused
path below, 256^2 / 128^2 tiles, CLS-only last layer, mean pooling) interleaved on one model; every result must be
bitwise equal to the first time its shape ran."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main(rounds=12):
    from convdr_amd.model.models import MSMarcoConfigDict, RobertaConfig
    torch.manual_seed(0)
    model = MSMarcoConfigDict["rdot_nll"].model_class(RobertaConfig(num_hidden_layers=2)).cuda().eval()
    rs = np.random.RandomState(0)
    shapes = []
    for B, L in ((256, 128), (3, 17), (64, 512), (1, 1), (400, 64), (40, 130), (700, 40), (9, 256)):
        lens = rs.randint(1, L + 1, size=B)
        lens[0] = L
        ids = rs.randint(3, 50000, size=(B, L)).astype(np.int64)
        ids[:, 0] = 0
        mask = (np.arange(L)[None, :] < lens[:, None]).astype(np.int64)
        ids[mask == 0] = 1
        shapes.append((torch.from_numpy(ids).cuda(), torch.from_numpy(mask).cuda()))
    bad = 0
    with torch.no_grad():
        ref = [model(i, m).clone() for i, m in shapes]
        for use_mean in (False, True):
            model.use_mean = use_mean
            if use_mean:
                ref = [model(i, m).clone() for i, m in shapes]
            for r in range(rounds):
                for j in rs.permutation(len(shapes)):
                    out = model(*shapes[j])
                    if not torch.equal(out, ref[j]):
                        bad += 1
                        print("use_mean=%s round %d shape %d %s differs: max abs %.3g" % (
                            use_mean, r, j, tuple(shapes[j][0].shape), (out - ref[j]).abs().max().item()), flush=True)
    print("differences:", bad)


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 12)
