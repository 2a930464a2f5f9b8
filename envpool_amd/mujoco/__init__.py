"""MuJoCo task families (only the gym family is on the MI355X path)."""
