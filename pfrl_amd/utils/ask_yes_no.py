"""Interactive yes / no prompt (reference pfrl/utils/ask_yes_no.py)."""


def ask_yes_no(question):
    """Keep asking until the answer starts like "yes" or "no"; returns True for yes."""
    answers = {"y": True, "ye": True, "yes": True, "n": False, "no": False}
    while True:
        choice = input("{} [y/N]: ".format(question)).lower()
        if choice in answers:
            return answers[choice]
