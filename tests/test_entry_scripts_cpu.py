"""The pure parts of the entry scripts (pit.txt): the challenger's score and the three
pit.txt line formats of pit.py:17,37-55, the win/draw tallies of test_pit.py:27-63, the weight report of
test_weights.py:8-12.  The expected strings are written out here from the reference's format (its spelling included)."""
import importlib.util
import io
import os

import numpy as np

from conftest import PKG


def load_script(name):
    spec = importlib.util.spec_from_file_location("entry_" + name, os.path.join(PKG, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_challenger_score_counts_draws_as_half_a_point_each():
    pit = load_script("pit")
    assert pit.challenger_score([1, 1, 0, None], 1) == 2.5 / 4.0
    assert pit.challenger_score([None, None], 1) == 0.5
    assert pit.challenger_score([0, 0, 0], 1) == 0.0
    assert pit.challenger_score([2, 3, 0, 1], 2) == 0.5                 # ids below the champion's snake count are its wins
    ids = [1] * 511 + [0] * 489                                         # 0.511 > 0.51: takes the title; 0.51 does not
    assert pit.challenger_score(ids, 1) == 511.0 / 1000.0
    assert pit.challenger_score([1] * 51 + [0] * 49, 1) == 0.51
    assert (pit.PIT_GAMES, pit.THRESHOLD, pit.HEIGHT, pit.WIDTH, pit.SNAKE_CNT) == (1000, 0.51, 11, 11, 2)


def test_pit_txt_lines_are_the_reference_format():
    pit = load_script("pit")
    assert pit.verdict_line("snake7", 0.5625) == \
        "snake7 beats the previouse champion. score = 0.5625. It is the new champion!\n"
    assert pit.verdict_line("snake8", 0.51) == "snake8 failed to beat the previouse champion. score = 0.51.\n"
    assert pit.verdict_line("snake9", 1.0 / 3.0) == \
        "snake9 failed to beat the previouse champion. score = " + str(1.0 / 3.0) + ".\n"


def test_test_pit_tally():
    tp = load_script("test_pit")
    assert tp.tally([0, None, 2, 3, 0, 1], 1) == (2, 3, 1)
    assert tp.tally([0, 1, None, 1], 1) == (1, 2, 1)
    assert (tp.HEIGHT, tp.WIDTH, tp.HEALTH_DEC, tp.GAMES) == (11, 11, 1, 300)


def test_weight_report_format():
    tw = load_script("test_weights")
    out = io.StringIO()
    w = [np.array([[1.0, -2.0], [0.5, 0.0]], np.float32), np.array([3.0], np.float32)]
    tw.report(w, out)
    assert out.getvalue() == ("(2, 2)\nMin weight: -2.0 Max weight: 1.0\nSum of squres (L2) 5.25\n\n"
                              "(1,)\nMin weight: 3.0 Max weight: 3.0\nSum of squres (L2) 9.0\n\n")


def test_train_settings_are_the_reference_defaults():
    tr = load_script("train")
    assert tr.SETTINGS == dict(game_board_height=11, game_board_width=11, number_of_snakes=4, self_play_games=256,
                               max_MCTS_depth=8, max_MCTS_breadth=128, initial_learning_rate=0.0001,
                               learning_rate_decay=0.98)
    assert tr.join_ranks() == (0, 1) or os.environ.get("WORLD_SIZE", "1") != "1"
