"""End to end on an MI355X: gz corpus -> readers -> native loader -> HIP training step -> device evaluation -> ranklist
(SURVEY.md §8a row T, §8f N1-N3), on a synthetic corpus small enough for seconds."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_train_validate_test_roundtrip(tmp_path):
    from prodsearch_amd import default_args, synth, trainer, corpus
    data_path, inp = synth.write_corpus(str(tmp_path / 'corpus'), 21, n_users=60, n_products=80, n_words=200)
    save = str(tmp_path / 'run')
    args = default_args(model_name='item_transformer', embedding_size=32, ff_size=64, heads=4, inter_layers=1,
                        batch_size=32, neg_per_pos=5, uprev_review_limit=5, subsampling_rate=1e-2, lr=0.01,
                        max_train_epoch=3, steps_per_checkpoint=20, has_valid=True, valid_candi_size=-1,
                        valid_batch_size=24, data_dir=data_path, input_train_dir=inp, save_dir=save, device='cuda',
                        dropout=0.1)
    np.random.seed(5)
    mrr, p1 = trainer.train(args)
    assert 0.0 < mrr <= 1.0 and 0.0 <= p1 <= 1.0
    assert os.path.exists(os.path.join(save, 'model_epoch_3.ckpt')) and os.path.exists(os.path.join(save, 'model_best.ckpt'))
    lines = open(os.path.join(save, args.rankfname)).read().splitlines()
    assert lines and all(len(ln.split(' ')) == 6 and ln.endswith('ReviewTransformer') for ln in lines)
    first = lines[0].split(' ')
    assert first[1] == 'Q0' and first[3] == '1' and first[2].startswith('B')
    # the checkpoint is the reference's dictionary and reloads to the same evaluation numbers
    ck = torch.load(os.path.join(save, 'model_best.ckpt'), map_location='cpu', weights_only=False)
    assert set(ck) == {'epoch', 'model', 'opt', 'optim'}
    gd = corpus.GlobalProdSearchData(args, data_path, inp)
    train_pd = corpus.ProdSearchData(args, inp, 'train', gd)
    test_pd = corpus.ProdSearchData(args, inp, 'test', gd)
    model, _ = trainer.create_model(args, gd, train_pd, os.path.join(save, 'model_best.ckpt'))
    mrr2, p12 = trainer.Trainer(args, model, None).test(args, gd, test_pd, 'again.ranklist')
    assert mrr2 == mrr and p12 == p1
    assert open(os.path.join(save, 'again.ranklist')).read().splitlines() == lines


def test_training_learns_on_a_tiny_corpus(tmp_path):
    """The loss the trainer logs falls over epochs and validation MRR beats the untrained model's."""
    from prodsearch_amd import default_args, synth, trainer, corpus, pyrandom
    data_path, inp = synth.write_corpus(str(tmp_path / 'c'), 22, n_users=50, n_products=40, n_words=150)
    args = default_args(model_name='item_transformer', embedding_size=32, ff_size=64, heads=4, inter_layers=1,
                        batch_size=64, neg_per_pos=5, uprev_review_limit=5, subsampling_rate=1e-1, lr=0.02,
                        max_train_epoch=6, steps_per_checkpoint=1000, has_valid=True, valid_candi_size=10,
                        data_dir=data_path, input_train_dir=inp, save_dir=str(tmp_path / 'r'), device='cuda', dropout=0.0)
    os.makedirs(args.save_dir)
    args.start_epoch = 0
    torch.manual_seed(1); pyrandom.seed(1); np.random.seed(1)
    gd = corpus.GlobalProdSearchData(args, data_path, inp)
    train_pd = corpus.ProdSearchData(args, inp, 'train', gd)
    valid_pd = corpus.ProdSearchData(args, inp, 'valid', gd)
    model, optim = trainer.create_model(args, gd, train_pd)
    tr = trainer.Trainer(args, model, optim)
    vds = corpus.ItemPVDataset(args, gd, valid_pd)
    before, _ = tr.validate(args, gd, vds)            # sampled candidates: model.test path
    tr.train(args, gd, train_pd, valid_pd)
    after, _ = tr.validate(args, gd, vds)
    assert after > before


@pytest.mark.parametrize('encoder,extra', [('pvc', dict()), ('pv', dict(train_pv_epoch=1, pv_window_size=4)),
                                           ('pvc', dict(use_user_emb=True, use_item_emb=True, do_seq_review_train=True,
                                                        do_seq_review_test=True, train_review_only=False))])
def test_review_transformer_roundtrip(tmp_path, encoder, extra):
    """review_transformer end to end: corpus -> ProdSearchDataset -> native loader (device-side review-word gather)
    -> ProductRanker step -> candidate evaluation -> ranklist; the best checkpoint reloads to the same numbers."""
    from prodsearch_amd import default_args, synth, trainer, corpus
    data_path, inp = synth.write_corpus(str(tmp_path / 'corpus'), 31, n_users=50, n_products=40, n_words=150)
    save = str(tmp_path / 'run')
    args = default_args(model_name='review_transformer', review_encoder_name=encoder, embedding_size=32, ff_size=64,
                        heads=4, inter_layers=1, batch_size=16, neg_per_pos=3, uprev_review_limit=3, iprev_review_limit=4,
                        review_word_limit=12, subsampling_rate=1e-2, lr=0.01, max_train_epoch=2, steps_per_checkpoint=5,
                        has_valid=True, valid_candi_size=8, candi_batch_size=8, test_candi_size=-1, valid_batch_size=6,
                        data_dir=data_path, input_train_dir=inp, save_dir=save, device='cuda', dropout=0.1, **extra)
    np.random.seed(7)
    mrr, p1 = trainer.train(args)
    assert 0.0 < mrr <= 1.0 and 0.0 <= p1 <= 1.0
    lines = open(os.path.join(save, args.rankfname)).read().splitlines()
    assert lines and all(len(ln.split(' ')) == 6 and ln.endswith('ReviewTransformer') for ln in lines)
    gd = corpus.GlobalProdSearchData(args, data_path, inp)
    train_pd = corpus.ProdSearchData(args, inp, 'train', gd)
    test_pd = corpus.ProdSearchData(args, inp, 'test', gd)
    model, _ = trainer.create_model(args, gd, train_pd, os.path.join(save, 'model_best.ckpt'))
    mrr2, p12 = trainer.Trainer(args, model, None).test(args, gd, test_pd, 'again.ranklist')
    assert mrr2 == mrr and p12 == p1
    assert open(os.path.join(save, 'again.ranklist')).read().splitlines() == lines


def test_review_transformer_learns(tmp_path):
    from prodsearch_amd import default_args, synth, trainer, corpus, pyrandom
    data_path, inp = synth.write_corpus(str(tmp_path / 'c'), 33, n_users=50, n_products=40, n_words=150)
    args = default_args(model_name='review_transformer', review_encoder_name='pvc', embedding_size=32, ff_size=64, heads=4,
                        inter_layers=1, batch_size=32, neg_per_pos=5, uprev_review_limit=4, iprev_review_limit=6,
                        review_word_limit=16, subsampling_rate=1e-1, lr=0.01, max_train_epoch=8, steps_per_checkpoint=1000,
                        has_valid=True, valid_candi_size=10, candi_batch_size=10, data_dir=data_path, input_train_dir=inp,
                        save_dir=str(tmp_path / 'r'), device='cuda', dropout=0.0, corrupt_rate=0.5)
    os.makedirs(args.save_dir)
    args.start_epoch = 0
    torch.manual_seed(1); pyrandom.seed(1); np.random.seed(1)
    gd = corpus.GlobalProdSearchData(args, data_path, inp)
    train_pd = corpus.ProdSearchData(args, inp, 'train', gd)
    valid_pd = corpus.ProdSearchData(args, inp, 'valid', gd)
    model, optim = trainer.create_model(args, gd, train_pd)
    tr = trainer.Trainer(args, model, optim)
    vds = corpus.ProdSearchDataset(args, gd, valid_pd)
    before, _ = tr.validate(args, gd, vds)
    tr.train(args, gd, train_pd, valid_pd)
    after, _ = tr.validate(args, gd, vds)
    assert after > before
