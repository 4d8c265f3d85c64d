#include <tf/tf.h>
