"""Test helper: write a tiny synthetic sentence-transformers model directory (the layout
SentenceTransformer(path) reads, reference rag/embedding.py:33) with safetensors weights."""
import json
import os

import numpy as np

WORDS = ("the quick brown fox jumps over lazy dog retrieval augmented generation embeds chunks of text "
         "vector store cosine similarity query answer context model paper cafe resume naive a b c d e").split()
PIECES = ["##s", "##ing", "##ed", "##al", "##ion", "##er", "##ly", "##e", "##a", "##t", "##n", "##r", "##i", "##o"]


def make_vocab():
    toks = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"] + sorted(set(WORDS)) + PIECES + list("abcdefghijklmnopqrstuvwxyz0123456789.,!?-'")
    seen, out = set(), []
    for t in toks:
        if t not in seen:
            seen.add(t); out.append(t)
    return out


def write_model_dir(path, *, pooling="mean", bert_prefix=True, hidden=64, layers=2, heads=4, ffn=128, max_pos=64, max_seq=48,
                    seed=0, sbert_lower=False, tok_lower=True, tokenizer_json=False, modules_json=True):
    """Returns (weights dict with HF BertModel names, config dict)."""
    from safetensors.numpy import save_file
    os.makedirs(os.path.join(path, "1_Pooling"), exist_ok=True)
    vocab = make_vocab()
    cfg = {"architectures": ["BertModel"], "model_type": "bert", "vocab_size": len(vocab), "hidden_size": hidden,
           "num_hidden_layers": layers, "num_attention_heads": heads, "intermediate_size": ffn,
           "max_position_embeddings": max_pos, "type_vocab_size": 2, "layer_norm_eps": 1e-12, "hidden_act": "gelu"}
    json.dump(cfg, open(os.path.join(path, "config.json"), "w"))
    json.dump({"max_seq_length": max_seq, "do_lower_case": sbert_lower}, open(os.path.join(path, "sentence_bert_config.json"), "w"))
    json.dump({"do_lower_case": tok_lower, "tokenizer_class": "BertTokenizer", "cls_token": "[CLS]", "sep_token": "[SEP]"},
              open(os.path.join(path, "tokenizer_config.json"), "w"))
    json.dump({"word_embedding_dimension": hidden, "pooling_mode_cls_token": pooling == "cls",
               "pooling_mode_mean_tokens": pooling == "mean", "pooling_mode_max_tokens": False,
               "pooling_mode_mean_sqrt_len_tokens": False}, open(os.path.join(path, "1_Pooling", "config.json"), "w"))
    if modules_json:
        json.dump([{"idx": 0, "name": "0", "path": "", "type": "sentence_transformers.models.Transformer"},
                   {"idx": 1, "name": "1", "path": "1_Pooling", "type": "sentence_transformers.models.Pooling"},
                   {"idx": 2, "name": "2", "path": "2_Normalize", "type": "sentence_transformers.models.Normalize"}],
                  open(os.path.join(path, "modules.json"), "w"))
    with open(os.path.join(path, "vocab.txt"), "w", encoding="utf-8") as fh:
        fh.write("\n".join(vocab) + "\n")
    if tokenizer_json:
        from rag.tokenizer import FastWordPieceTokenizer
        FastWordPieceTokenizer.from_vocab({t: i for i, t in enumerate(vocab)}, lower=tok_lower)._tok.save(os.path.join(path, "tokenizer.json"))
    rng = np.random.default_rng(seed)
    h, f = hidden, ffn
    w = {"embeddings.word_embeddings.weight": (len(vocab), h), "embeddings.position_embeddings.weight": (max_pos, h),
         "embeddings.token_type_embeddings.weight": (2, h), "embeddings.LayerNorm.weight": (h,), "embeddings.LayerNorm.bias": (h,)}
    for i in range(layers):
        p = f"encoder.layer.{i}."
        w.update({p + "attention.self.query.weight": (h, h), p + "attention.self.query.bias": (h,),
                  p + "attention.self.key.weight": (h, h), p + "attention.self.key.bias": (h,),
                  p + "attention.self.value.weight": (h, h), p + "attention.self.value.bias": (h,),
                  p + "attention.output.dense.weight": (h, h), p + "attention.output.dense.bias": (h,),
                  p + "attention.output.LayerNorm.weight": (h,), p + "attention.output.LayerNorm.bias": (h,),
                  p + "intermediate.dense.weight": (f, h), p + "intermediate.dense.bias": (f,),
                  p + "output.dense.weight": (h, f), p + "output.dense.bias": (h,),
                  p + "output.LayerNorm.weight": (h,), p + "output.LayerNorm.bias": (h,)})
    weights = {}
    for name, shp in w.items():
        if name.endswith("LayerNorm.weight"):
            a = 1.0 + 0.05 * rng.standard_normal(shp)
        elif name.endswith(".bias"):
            a = 0.02 * rng.standard_normal(shp)
        else:
            a = 0.08 * rng.standard_normal(shp)
        weights[name] = a.astype(np.float32)
    disk = {("bert." + k if bert_prefix else k): v for k, v in weights.items()}
    disk[("bert." if bert_prefix else "") + "pooler.dense.weight"] = np.zeros((h, h), dtype=np.float32)   # present in real checkpoints, unused
    save_file(disk, os.path.join(path, "model.safetensors"))
    return weights, cfg
