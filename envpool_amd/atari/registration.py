"""Atari task ids (mirror of envpool/atari/registration.py:17-37).

The reference registers `<Game>-v5` for every ROM file found under
`<base_path>/atari/roms`.  ROMs are not redistributable and not part of this repository:
when that directory exists its content is used like in the reference, otherwise the ids of
the ROM set ale-py 0.11 ships are registered so that `make("Pong-v5", ...)` resolves and
fails with a clear message about the missing ROM / emulator plugin instead of an unknown id.
"""
import os

from envpool_amd.registration import base_path, register

_ALE_GAMES = """adventure air_raid alien amidar assault asterix asteroids atlantis atlantis2
backgammon bank_heist basic_math battle_zone beam_rider berzerk blackjack bowling boxing
breakout carnival casino centipede chopper_command crazy_climber crossbow darkchambers
defender demon_attack donkey_kong double_dunk earthworld elevator_action enduro entombed et
fishing_derby flag_capture freeway frogger frostbite galaxian gopher gravitar hangman
haunted_house hero human_cannonball ice_hockey jamesbond journey_escape kaboom kangaroo
keystone_kapers king_kong klax koolaid krull kung_fu_master laser_gates lost_luggage
mario_bros miniature_golf montezuma_revenge mr_do ms_pacman name_this_game othello pacman
phoenix pitfall pitfall2 pong pooyan private_eye qbert riverraid road_runner robotank
seaquest sir_lancelot skiing solaris space_invaders space_war star_gunner superman surround
tennis tetris tic_tac_toe_3d time_pilot trondead turmoil tutankham up_n_down venture
video_checkers video_chess video_cube video_pinball wizard_of_wor word_zapper yars_revenge
zaxxon""".split()

atari_rom_path = os.path.join(base_path, "atari", "roms")
if os.path.isdir(atari_rom_path):
    atari_game_list = sorted(i.replace(".bin", "") for i in os.listdir(atari_rom_path)
                             if i.endswith(".bin"))
else:
    atari_game_list = sorted(_ALE_GAMES)

for game in atari_game_list:
    name = "".join(g.capitalize() for g in game.split("_"))
    register(
        task_id=name + "-v5",
        import_path="envpool_amd.atari",
        spec_cls="AtariEnvSpec",
        dm_cls="AtariDMEnvPool",
        gymnasium_cls="AtariGymnasiumEnvPool",
        task=game,
        max_episode_steps=27000,
    )
