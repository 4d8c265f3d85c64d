#include <ros/time.h>
