"""CPU: pin the oracle's random streams (MT19937 == torch CPU generator; Philox4x32-10 == Random123 vectors)."""
import numpy as np

import golden_util as G


def test_mt19937_matches_torch_cpu_generator(oracle):
    """utils/random_generator.py:76-114: per-env torch CPU generator, float32 = (u32 & 0xFFFFFF) * 2^-24."""
    data = np.load(G.golden_path('mt19937_torch.npz'))
    seeds, want = data['seeds'], data['draws']
    state, index = oracle.mt19937_seed(seeds)
    n = want.shape[1]
    first = oracle.mt19937_generate(state, index, 3, 6)          # generate(B, 3, (2, 3))
    second = oracle.mt19937_generate(state, index, 5, 3)         # generate(B, 5, (3,))
    rest = oracle.mt19937_generate(state, index, 1, n - 33)      # crosses two 624-word twists
    got = np.concatenate([first.transpose(1, 0, 2).reshape(len(seeds), -1), second.transpose(1, 0, 2).reshape(len(seeds), -1),
                          rest[0]], axis=1)
    G.assert_same(got, want, 'MT19937 float stream')


def test_mt19937_published_first_outputs(oracle):
    """init_genrand(5489) first outputs of the published mt19937ar.c."""
    state, index = oracle.mt19937_seed(np.array([5489], np.int32))
    out = oracle.mt19937_generate(state, index, 1, 3)[0, 0]
    want = np.array([3499211612, 581869302, 3890346734], np.uint32)
    G.assert_same(out, (want & 0xFFFFFF).astype(np.float32) / np.float32(16777216.0), 'genrand_int32 low 24 bits')


def test_philox_known_answers(oracle):
    """Random123 kat_vectors: philox4x32-10."""
    assert oracle.philox4x32_10([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert oracle.philox4x32_10([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert oracle.philox4x32_10([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344],
                                [0xa4093822, 0x299f31d0]) == [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_wildfire_philox_stream_definition(oracle):
    """include/frz.h FRZ_RNG_PHILOX: a 128-bit block is five 24-bit uniforms; draw u = field u % 5 of block u // 5."""
    from free_range_zoo_amd import _capi
    cfg = _capi.frz_wildfire_cfg()
    cfg.parallel_envs, cfg.grid_height, cfg.grid_width, cfg.num_agents = 2, 2, 3, 3
    seeds, moves = np.array([7, 11], np.int32), np.array([3, 5], np.int32)
    field, agent = oracle.wildfire_philox_randomness(cfg, seeds, moves)

    def draw(b, u):
        blk = oracle.philox4x32_10([u // 5, int(moves[b]), 0, 0], [int(seeds[b]), 0x46525A00])
        big = blk[0] | (blk[1] << 32) | (blk[2] << 64) | (blk[3] << 96)
        return np.float32((big >> (24 * (u % 5))) & 0xFFFFFF) / np.float32(16777216.0)

    for b in range(2):
        for e in range(3):
            for c in range(6):
                assert field[e, b, c] == draw(b, e * 6 + c)
        for e in range(5):
            for a in range(3):
                assert agent[e, b, a] == draw(b, 18 + e * 3 + a)
    assert field.min() >= 0.0 and field.max() < 1.0
