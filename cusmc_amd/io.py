"""writeOutput() -- the CSV side effects of run() (src/io.cpp:7-43): `y_t.csv` (header `y`, one
row per time step, every value followed by a comma) and `x_t_N<p>.csv` (header `w,x`; per time
step the FIRST particle's weight w_t[i][0], then particle p's d coordinates)."""
import os


def _fmt(v):
    return "%g" % v  # std::ofstream default formatting: 6 significant digits


def writeOutput(y_t, w_t, post_x_t, N, d, timeSteps, p, directory="."):
    del N
    with open(os.path.join(directory, "y_t.csv"), "w") as fy, \
            open(os.path.join(directory, "x_t_N%d.csv" % p), "w") as fx:
        fy.write("y\n")
        fx.write("w,x\n")
        for i in range(timeSteps):
            fx.write(_fmt(w_t[i][0]))
            for j in range(d):
                fy.write(_fmt(y_t[i][j]) + ",")
                fx.write("," + _fmt(post_x_t[i][p][j]))
            fy.write("\n")
            fx.write("\n")
