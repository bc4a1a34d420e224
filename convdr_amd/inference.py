"""Query-encode loop of the retrieval driver on MI355X.

Mirrors /root/reference/drivers/run_convdr_inference.py:116-154 (``evaluate``), same signature and return value:
the eval set is walked in order in batches of ``per_gpu_eval_batch_size * max(1, n_gpu)``, every batch is encoded by
``model(concat_ids, concat_id_mask)`` (the HIP encoder behind ``model.models``), and the function returns
``(embedding float32 [N, 768] numpy, embedding2id list of query ids, raw_sequences list of history utterances)``.

What changed for the GPU (results identical): the reference copies every batch's embeddings to the host synchronously
(``embs.detach().cpu().numpy()`` per batch, :145); here the batches stay on the device and come back in ONE copy after the
loop, so the host keeps enqueueing while the GPU encodes (query batches are 4 x <= 510 tokens: launch-bound).
"""
import random

import numpy as np
import torch
from torch.utils.data import DataLoader, SequentialSampler


def set_seed(args):
    """utils/util.py:233-238."""
    random.seed(args.seed)
    np.random.seed(args.seed)
    torch.manual_seed(args.seed)
    if getattr(args, "n_gpu", 0) > 0 and torch.cuda.is_available():
        torch.cuda.manual_seed_all(args.seed)


def evaluate(args, eval_dataset, model, logger=None):
    args.eval_batch_size = args.per_gpu_eval_batch_size * max(1, args.n_gpu)
    eval_sampler = SequentialSampler(eval_dataset)
    eval_dataloader = DataLoader(eval_dataset, sampler=eval_sampler, batch_size=args.eval_batch_size,
                                 collate_fn=eval_dataset.get_collate_fn(args, "inference"))
    if logger is not None:
        logger.info("***** Running evaluation *****")
        logger.info("  Num examples = %d", len(eval_dataset))
        logger.info("  Instantaneous batch size per GPU = %d", args.per_gpu_eval_batch_size)
    model.zero_grad()
    set_seed(args)  # the reference re-seeds here (:132-133)
    embedding, embedding2id, raw_sequences = [], [], []
    model.eval()
    import inspect
    fwd = (model.module if hasattr(model, "module") else model).forward
    takes_lens = "seq_lens" in inspect.signature(fwd).parameters      # (BiEncoder.forward has no such extension)
    for batch in eval_dataloader:
        qids = batch["qid"]
        ids, id_mask = (ele.to(args.device, non_blocking=True) for ele in [batch["concat_ids"], batch["concat_id_mask"]])
        with torch.no_grad():
            # the collate function right-pads (utils/util.py:163-185): the host knows the lengths, no device round trip
            lens = batch["concat_id_mask"].sum(1).numpy().astype(np.int32)
            embs = model(ids, id_mask, seq_lens=lens) if takes_lens else model(ids, id_mask)
        embedding.append(embs.detach())
        embedding2id.extend(qids)
        raw_sequences.extend(batch["history_utterances"])
    if not embedding:
        return np.zeros((0, 768), np.float32), embedding2id, raw_sequences
    embedding = torch.cat(embedding, 0).cpu().numpy()
    from .train import check_status
    check_status(model)      # token ids outside the embedding table: IndexError like the reference's lookup (models.py:141)
    return embedding, embedding2id, raw_sequences
