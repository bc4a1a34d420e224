"""Oracle: the query-encode loop (CPU).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates /root/reference/drivers/run_convdr_inference.py:116-154 (``evaluate``): the eval set is walked in order
(SequentialSampler) in batches of ``per_gpu_eval_batch_size * max(1, n_gpu)``; every batch goes through
``model(concat_ids, concat_id_mask)`` (= NLL.forward -> query_emb, models.py:60-62) under ``model.eval()`` /
``torch.no_grad()``; the embeddings are concatenated on the host in batch order, the query ids and the raw history
utterances are collected in the same order.  Pinned by tests/golden/evaluate.npz (the reference's own function run on a
stub dataset, tests/golden/make_golden.py:gen_evaluate).
"""
import numpy as np
import torch

from . import encoder as OE


def evaluate(sd, ids, mask, qids, history_utterances, batch_size, *, num_layers, num_heads):
    """-> (embedding float32 [N, 768], embedding2id list, raw_sequences list), as the reference returns them."""
    embedding, embedding2id, raw_sequences = [], [], []
    for s in range(0, len(qids), batch_size):                       # SequentialSampler + DataLoader(batch_size)  :118-123
        e = slice(s, s + batch_size)
        with torch.no_grad():                                        # :143-144
            embs = OE.rdot_nll_emb(sd, torch.as_tensor(ids[e]), torch.as_tensor(mask[e]), num_layers=num_layers,
                                   num_heads=num_heads)
        embedding.append(embs.detach().cpu().numpy())               # :145-146
        embedding2id.extend(qids[e])                                 # :147-148
        raw_sequences.extend(history_utterances[e])                  # :150-151
    return np.concatenate(embedding, axis=0), embedding2id, raw_sequences   # :153-154
